"""K3 parity: fused MFMA CNN forward (through the C ABI) vs golden vectors recorded from the reference's torch
module (fp32, real checkpoint) and vs the CPU oracle.  Tolerance: 1e-4 relative (BASELINE.json north_star)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def blob():
    return np.fromfile(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")


@pytest.fixture(scope="module", params=["wino", "split"])
def eng(blob, request):
    """one engine per K3 kernel (az_net_set_kernel): every test that takes `eng` runs on the Winograd fp32-MFMA kernel
    and on the split-fp16 kernel, at the same tolerance"""
    from alphazero_amd.engine import SelfPlayEngine
    e = SelfPlayEngine("Connect4", n_games=64, n_playout=50, c_puct=1.25, evaluator="cnn", flip="stream",
                       reuse_tree=False)
    e.load_net(blob, kernel=request.param)
    assert e.k3_kernel == request.param
    yield e
    e.close()


def _packed(cells):
    return np.array([orc.Env.from_cells("Connect4", c).packed() for c in cells])


CKPTS = ["AZ_Connect4_CNN_current", "AZ_Connect4_CNN_best", "AZ2_Connect4_CNN_current", "AZ2_Connect4_CNN_best"]


@pytest.mark.parametrize("kernel", ["wino", "split", "direct"])
def test_k3_all_four_reference_checkpoints(kernel):
    """every checkpoint the reference ships (params/*.pt re-exported as blobs), every K3 kernel: outputs vs the vectors
    recorded from the reference's torch module at the 1e-4 contract, and the error against the SAME module run in
    float64 next to torch-float32's own (tests/golden/make_fixtures_cnn4.py): each kernel must be as close to the exact
    network as fp32 arithmetic is (within 1.5 x of torch's mean error, 2 x of its max, per checkpoint)."""
    from alphazero_amd.engine import SelfPlayEngine
    d = np.load(os.path.join(G, "cnn4.npz"))
    packed = _packed(d["cells"])
    e = SelfPlayEngine("Connect4", n_games=8, n_playout=8, evaluator="cnn")
    for name in CKPTS:
        b = np.fromfile(os.path.join(G, "weights", name + ".f32"), dtype="<f4")
        e.load_net(b, kernel=kernel)          # also the hot-reload path: four loads on one live engine
        assert e.k3_kernel == kernel
        logp, logv, probs, value = e.net_forward(packed)
        np.testing.assert_allclose(logp, d[name + "/logp"], rtol=1e-4, atol=1e-5, err_msg=name)
        np.testing.assert_allclose(logv, d[name + "/logv"], rtol=1e-4, atol=1e-5, err_msg=name)
        np.testing.assert_allclose(probs, d[name + "/probs"], rtol=1e-4, atol=1e-7, err_msg=name)
        np.testing.assert_allclose(value, d[name + "/value"], rtol=1e-4, atol=1e-5, err_msg=name)
        ek = np.abs(logp - d[name + "/logp64"])
        et = np.abs(d[name + "/logp"] - d[name + "/logp64"])
        print("%-6s %-26s |dlogp| vs float64: kernel max %.2e mean %.2e | torch f32 max %.2e mean %.2e"
              % (kernel, name, ek.max(), ek.mean(), et.max(), et.mean()))
        assert ek.mean() <= 1.5 * et.mean() and ek.max() <= 2.0 * et.max(), (kernel, name)
    e.close()


@pytest.mark.parametrize("kernel", ["wino", "split"])
def test_k3_range_stress_variants(kernel, blob):
    """weights rescaled by 2^+-k per layer (activations up to ~2^20 and down to ~2^-20: far outside what a fixed-scale
    fp16 split could hold) and perturbed BatchNorm statistics (tests/helpers.py: stress_blob; golden outputs from the
    reference's torch module fed the same transformed weights).  These networks amplify rounding differently, so the
    bar is stated against the float64 forward: the kernel's error must stay within 3 x torch-float32's own (mean) and
    4 x (max), and nothing may be non-finite."""
    from helpers import STRESS_VARIANTS, stress_blob
    from alphazero_amd.engine import SelfPlayEngine
    d = np.load(os.path.join(G, "cnn4.npz"))
    packed = _packed(d["cells"])
    e = SelfPlayEngine("Connect4", n_games=8, n_playout=8, evaluator="cnn")
    for var in STRESS_VARIANTS:
        e.load_net(stress_blob(blob, var), kernel=kernel)
        logp, logv, probs, value = e.net_forward(packed)
        key = "stress_%s/" % var
        assert np.isfinite(logp).all() and np.isfinite(value).all(), var
        ek, et = np.abs(logp - d[key + "logp64"]), np.abs(d[key + "logp"] - d[key + "logp64"])
        ev, tv = np.abs(logv - d[key + "logv64"]), np.abs(d[key + "logv"] - d[key + "logv64"])
        print("%-6s %-10s |dlogp| vs float64: kernel max %.2e mean %.2e | torch f32 max %.2e mean %.2e"
              % (kernel, var, ek.max(), ek.mean(), et.max(), et.mean()))
        assert ek.mean() <= 3 * et.mean() + 1e-7 and ek.max() <= 4 * et.max() + 1e-6, var
        assert ev.mean() <= 3 * tv.mean() + 1e-7 and ev.max() <= 4 * tv.max() + 1e-6, var
        np.testing.assert_allclose(probs.sum(1), 1.0, rtol=1e-5)
    e.close()


def test_net_load_refuses_non_finite_weights(blob):
    from alphazero_amd._abi import AzError
    from alphazero_amd.engine import SelfPlayEngine
    e = SelfPlayEngine("Connect4", n_games=4, n_playout=4, evaluator="cnn")
    bad = blob.copy()
    bad[12345] = np.inf
    with pytest.raises(AzError):
        e.load_net(bad)
    e.close()


def test_cnn_vs_reference_golden(eng):
    d = np.load(os.path.join(G, "cnn.npz"))
    envs = [orc.Env.from_cells("Connect4", c) for c in d["cells"]]
    packed = np.array([e.packed() for e in envs])
    logp, logv, probs, value = eng.net_forward(packed)
    np.testing.assert_allclose(logp, d["logp"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logv, d["logv"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(probs, d["probs"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(value, d["value"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 11, 127, 1000])
def test_cnn_ragged_batches_vs_oracle(eng, blob, n):
    """batch sizes that do not fill the 5-sample (Winograd kernel) or 3-sample (direct kernel) workgroup tile; random reachable positions."""
    rng = np.random.default_rng(n)
    envs = []
    while len(envs) < n:
        e = orc.Env("Connect4")
        for _ in range(int(rng.integers(0, 30))):
            if e.done():
                break
            e.step(int(rng.choice(e.valid_move())))
        if not e.done():
            envs.append(e)
    packed = np.array([e.packed() for e in envs])
    logp, logv, probs, value = eng.net_forward(packed)
    m = min(n, 40)
    onet = orc.CNN(blob)
    op, ov = onet.predict(np.array([e.current_state()[0] for e in envs[:m]]))
    np.testing.assert_allclose(probs[:m], op, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(value[:m], ov, rtol=1e-4, atol=1e-5)
    # batch-composition independence: the same position gives the same bits wherever it sits in the batch
    perm = rng.permutation(n)
    _, _, probs2, value2 = eng.net_forward(packed[perm])
    assert np.array_equal(probs2, probs[perm]) and np.array_equal(value2, value[perm])
    np.testing.assert_allclose(probs.sum(1), 1.0, rtol=1e-5)


def test_split_forms_deliver_the_same_bits(blob):
    """The split-fp16 K3 evaluates launches of up to 64 leaves with two positions per workgroup and larger ones with three
    (az_cnn_split.hip, cnns_launch): the same position must come out with the same bits from either form, or a launch's
    size -- and with the evaluation cache, the history of a table -- would show in a result."""
    from alphazero_amd.engine import SelfPlayEngine
    rng = np.random.default_rng(77)
    envs = []
    while len(envs) < 300:
        e = orc.Env("Connect4")
        for _ in range(int(rng.integers(0, 34))):
            if e.done():
                break
            e.step(int(rng.choice(e.valid_move())))
        if not e.done():
            envs.append(e)
    packed = np.array([e.packed() for e in envs])
    eng = SelfPlayEngine("Connect4", n_games=1, n_playout=1, evaluator="cnn")
    eng.load_net(blob, kernel="split")
    big = eng.net_forward(packed)                      # 300 leaves: three positions per workgroup
    for lo, n in ((0, 1), (1, 2), (3, 7), (10, 64), (200, 33)):
        small = eng.net_forward(packed[lo:lo + n])     # <= 64 leaves: two positions per workgroup
        for a, b in zip(small, big):
            assert np.array_equal(a.view(np.uint32), b[lo:lo + n].view(np.uint32)), (lo, n)
    eng.close()


def test_search_with_real_net_move_agreement(eng):
    """BASELINE config 1 head: S=500, eval mode, fresh tree per ply, recorded flips, real checkpoint.
    The fp32 MFMA forward and torch's CPU forward differ in the last bits, so visit counts are not expected to be
    bit-identical; the stated metric is move agreement + visit L1 distance (north_star)."""
    from alphazero_amd.engine import SelfPlayEngine
    d = np.load(os.path.join(G, "realnet_search.npz"))
    S = int(d["S"])
    plies = len(d["chosen"])
    e = SelfPlayEngine("Connect4", n_games=1, n_playout=S, c_puct=1.25, evaluator="cnn", flip="stream",
                       reuse_tree=False)
    e.load_net(eng._keep, kernel=eng.k3_kernel)
    e.set_flip_stream(d["flips"].reshape(1, -1))
    agree, l1 = 0, []
    for ply in range(plies):
        e.search()
        rc = e.root_children()
        k = int(rc["n"][0])
        ref_acts = [a for a in d["actions"][ply] if a >= 0]
        assert list(rc["actions"][0][:k]) == ref_acts
        vis = rc["visits"][0][:k]
        assert vis.sum() == S - 1 and rc["root_visits"][0] == S
        l1.append(int(np.abs(vis - d["visits"][ply][:k]).sum()))
        a, _ = e.choose_moves(temp=0.0)
        agree += int(a[0] == d["chosen"][ply])
        e.advance(np.array([d["chosen"][ply]], dtype=np.int32))  # follow the reference line
    print("move agreement %d/%d, visit L1 per ply %s (S=%d)" % (agree, plies, l1, S))
    assert agree >= plies - 1
    assert max(l1) <= 0.1 * S
    e.close()


def test_search_result_independent_of_cohorts(blob):
    """az_config.n_cohorts only changes how the select -> CNN -> backup chains are issued (slot ranges on their own
    HIP streams); games are independent, so every slot's tree must come out identical for 1, 2, 3 and 4 cohorts."""
    from alphazero_amd.engine import SelfPlayEngine
    rng = np.random.default_rng(3)
    Gn, S = 96, 48
    flips = rng.integers(0, 2, size=(Gn, S * 3)).astype(np.uint8)
    roots = []
    for g in range(Gn):
        e = orc.Env("Connect4")
        for _ in range(int(rng.integers(0, 6))):
            e.step(int(rng.choice(e.valid_move())))
        roots.append(e.packed())
    roots = np.array(roots)
    ref = None
    for C in (1, 2, 3, 4):
        eng = SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, c_puct=1.25, evaluator="cnn", flip="stream",
                             reuse_tree=True, n_cohorts=C)
        assert eng.n_cohorts == C
        eng.load_net(blob)
        eng.set_flip_stream(flips)
        eng.set_roots(roots)
        out = []
        for ply in range(3):
            eng.search()
            rc = eng.root_children()
            a, pi = eng.choose_moves(temp=1e-3)
            out.append((rc["actions"].copy(), rc["visits"].copy(), rc["q"].view(np.uint32).copy(), a.copy()))
            eng.advance(a)
        eng.close()
        if ref is None:
            ref = out
        else:
            for x, y in zip(ref, out):
                for u, v in zip(x, y):
                    assert np.array_equal(u, v), C


def test_invariants_full_size_c4_cnn(blob):
    """BASELINE configs[2] shape (2048 Connect4 games x 500 playouts, real net, 4 cohorts) through size-independent
    properties: root N = S, sum(child N) = S - 1; after prune_root(a) + S more playouts sum(child N) = n_a - 1 + S;
    2048 identical games (eval mode, never-flip, first-max) give 2048 bit-identical trees wherever their leaves sit in
    the CNN batches (workgroup, tile column, cohort); every leaf either reached the network or was terminal."""
    from alphazero_amd.engine import SelfPlayEngine
    Gn, S = 2048, 500
    eng = SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, c_puct=1.25, evaluator="cnn", flip="never",
                         reuse_tree=True)
    assert eng.n_cohorts == 4
    eng.load_net(blob)
    eng.search()
    rc = eng.root_children()
    assert (rc["root_visits"] == S).all() and (rc["n"] == 7).all()
    assert (rc["visits"].sum(1) == S - 1).all()
    for k in ("actions", "visits"):
        assert (rc[k] == rc[k][0]).all(), k
    assert (rc["q"].view(np.uint32) == rc["q"].view(np.uint32)[0]).all()
    assert (rc["prior"].view(np.uint32) == rc["prior"].view(np.uint32)[0]).all()
    a, pi = eng.choose_moves(temp=0.0)
    assert (a == a[0]).all()
    n_a = rc["visits"][0][list(rc["actions"][0]).index(int(a[0]))]
    eng.advance(a)
    eng.search()
    rc2 = eng.root_children()
    assert (rc2["root_visits"] == n_a + S).all()
    assert (rc2["visits"].sum(1) == n_a - 1 + S).all()
    assert (rc2["visits"] == rc2["visits"][0]).all()
    st = eng.stats()
    assert st["playouts"] == 2 * Gn * S and st["leaf_evals"] + st["cache_hits"] + st["terminal_hits"] == st["playouts"]
    eng.close()


@pytest.mark.parametrize("noise", [0.0, 0.7])
def test_rolling_selfplay_equals_lockstep(blob, noise):
    """az_selfplay_rolling (no lockstep plies: terminal playouts finished in place, ply ends inside the tree step) must
    produce, game id by game id, exactly the trajectories of the lockstep az_selfplay: every game runs its S playouts per
    move sequentially on its own counter-keyed random streams in both, only the batching differs."""
    from alphazero_amd.engine import SelfPlayEngine
    Gn, S, N = 96, 40, 250

    def run(rolling):
        eng = SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, c_puct=1.25, evaluator="cnn", flip="random",
                             reuse_tree=True, dirichlet_alpha=noise, rng_seed=5, move="sample", n_cohorts=3)
        eng.load_net(blob)
        n = eng.selfplay_rolling(N, temp=1.0, first_n_steps=6) if rolling else eng.selfplay(N, temp=1.0, first_n_steps=6)
        t = eng.traj_fetch()
        st = eng.stats()
        eng.close()
        return n, t, st

    n1, t1, s1 = run(False)
    n2, t2, s2 = run(True)
    assert n1 == n2 == N

    def by_game(t):
        out = {}
        order = np.argsort(t["game_id"], kind="stable")
        for i in order:
            out.setdefault(int(t["game_id"][i]), []).append(i)
        return out

    g1, g2 = by_game(t1), by_game(t2)
    assert sorted(g1) == sorted(g2) == list(range(N))
    for gid in g1:
        i1, i2 = np.array(g1[gid]), np.array(g2[gid])
        assert len(i1) == len(i2), gid
        for k in ("states", "action", "player", "done"):
            assert np.array_equal(t1[k][i1], t2[k][i2]), (gid, k)
        assert np.array_equal(t1["pi"][i1].view(np.uint32), t2["pi"][i2].view(np.uint32)), gid
        assert np.array_equal(t1["z"][i1], t2["z"][i2]), gid
    for k in ("playouts", "terminal_hits", "plies", "games_finished", "levels", "children"):
        assert s1[k] == s2[k], (k, s1[k], s2[k])
    # which non-terminal leaves the evaluation cache answers depends on the order the games reach them in; their number does not
    assert s1["leaf_evals"] + s1["cache_hits"] == s2["leaf_evals"] + s2["cache_hits"] and s2["cache_hits"] > 0


def test_split_kernel_next_to_winograd_kernel(blob):
    """the two product-grade kernels on the same positions: outputs within the contract of each other, a position's
    result independent of its neighbours in the batch (the split kernel's per-position range scale included), and a
    search driven by the split kernel keeps the tree invariants."""
    from alphazero_amd.engine import SelfPlayEngine
    d = np.load(os.path.join(G, "cnn.npz"))
    packed = _packed(d["cells"])
    out = {}
    for k in ("wino", "split"):
        e = SelfPlayEngine("Connect4", n_games=64, n_playout=50, c_puct=1.25, evaluator="cnn", flip="never", reuse_tree=False)
        e.load_net(blob, kernel=k)
        out[k] = e.net_forward(packed)
        if k == "split":
            for n in (1, 2, 3, 4, 5, 7, 9, 130):
                lp, lv, pr, va = e.net_forward(packed[:n])
                assert np.array_equal(lp, out[k][0][:n]) and np.array_equal(va, out[k][3][:n])
            e.search()
            rc = e.root_children()
            assert (rc["root_visits"] == 50).all() and (rc["visits"].sum(1) == 49).all()
        e.close()
    np.testing.assert_allclose(out["split"][0], out["wino"][0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["split"][3], out["wino"][3], rtol=1e-4, atol=1e-5)
