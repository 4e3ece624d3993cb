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


@pytest.fixture(scope="module")
def eng(blob):
    from alphazero_amd.engine import SelfPlayEngine
    e = SelfPlayEngine("Connect4", n_games=64, n_playout=50, c_puct=1.25, evaluator="cnn", flip="stream",
                       reuse_tree=False)
    e.load_net(blob)
    yield e
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
    e.load_net(eng._keep)
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
    assert st["playouts"] == 2 * Gn * S and st["leaf_evals"] + st["terminal_hits"] == st["playouts"]
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
    for k in ("playouts", "leaf_evals", "terminal_hits", "plies", "games_finished", "levels", "children"):
        assert s1[k] == s2[k], (k, s1[k], s2[k])


@pytest.fixture()
def split_eng(blob):
    """an engine whose network is evaluated by the experimental split-fp16 kernel (az_cnn_split.hip, AZ_K3_SPLIT=1)"""
    from alphazero_amd.engine import SelfPlayEngine
    e = SelfPlayEngine("Connect4", n_games=64, n_playout=50, c_puct=1.25, evaluator="cnn", flip="never", reuse_tree=False)
    os.environ["AZ_K3_SPLIT"] = "1"
    try:
        e.load_net(blob)
    finally:
        del os.environ["AZ_K3_SPLIT"]
    yield e
    e.close()


def test_split_fp16_kernel_vs_reference_golden(split_eng, eng, blob):
    """the opt-in kernel (fp32 products split over fp16 MFMA) against the vectors recorded from the reference, at the
    SAME tolerance as the product kernel, ragged batches included (4 positions per workgroup), and next to the product
    kernel's own outputs; a search driven by it finds the same trees' visit totals."""
    d = np.load(os.path.join(G, "cnn.npz"))
    envs = [orc.Env.from_cells("Connect4", c) for c in d["cells"]]
    packed = np.array([e.packed() for e in envs])
    logp, logv, probs, value = split_eng.net_forward(packed)
    np.testing.assert_allclose(logp, d["logp"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logv, d["logv"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(probs, d["probs"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(value, d["value"], rtol=1e-4, atol=1e-5)
    wl, _, wp, wv = eng.net_forward(packed)
    print("split vs Winograd fp32 kernel: max |dlogp| %.2e, max |dvalue| %.2e" %
          (np.abs(logp - wl).max(), np.abs(value - wv).max()))
    for n in (1, 2, 3, 4, 5, 7, 9, 130):
        lp, lv, pr, va = split_eng.net_forward(packed[:n])
        assert np.array_equal(lp, logp[:n]) and np.array_equal(va, value[:n])   # batch-composition independence
    split_eng.search()
    rc = split_eng.root_children()
    assert (rc["root_visits"] == 50).all() and (rc["visits"].sum(1) == 49).all()
