"""K2/K4 parity: PUCT, batched tree search, move selection -- HIP (through the C ABI) vs golden vectors recorded
from the reference and vs the CPU oracle.  Visit counts, actions, Q and prior bit patterns: bit-exact."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def E():
    from alphazero_amd import engine
    return engine


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(G, "fixtures.json")) as f:
        return json.load(f)


def test_puct_bit_exact(E):
    d = np.load(os.path.join(G, "puct.npz"))
    for c in np.unique(d["c_init"]):
        m = d["c_init"] == c
        Q = np.where(d["n"][m] > 0, d["Q"][m], np.float32(0))
        got = E.puct_batch(float(c), 500, d["Np"][m], d["n"][m], Q, d["P"][m])
        assert np.array_equal(got.view(np.uint32), d["score_bits"][m])


def _start_env(m):
    if m["game"] == "NBTTT":
        return orc.Env.nbttt_open(*m["opening"])
    return orc.Env(m["game"])


def _run_fixture_case(E, name, m, d, split):
    game = m["game"]
    evaluator = "host" if split else m["kind"]
    eng = E.SelfPlayEngine(game, n_games=1, n_playout=m["S"], c_puct=m["c_puct"], evaluator=evaluator,
                           eval_seed=m["seed"], flip="stream" if m["has_flips"] else "never", reuse_tree=m["reuse"])
    if m["has_flips"]:
        eng.set_flip_stream(d[name + "__flips"].reshape(1, -1))
    env = _start_env(m)
    eng.set_roots(env.packed().reshape(1, -1))
    A = eng.A
    plies = len(d[name + "__action"])
    for ply in range(plies):
        first_n, temp = m["first_n"], m["temp"]
        temperature = (1e-3 if ply >= first_n else temp) if first_n is not None else temp
        if split:
            for _ in range(m["S"]):
                ls, fl = eng.select()
                probs = np.zeros((1, A), np.float32)
                vals = np.zeros(1, np.float32)
                if not fl[0] & 1:
                    if m["kind"] == "uniform":
                        probs[0, :] = np.float32(1.0) / np.float32(A)
                    else:
                        probs[0], vals[0] = orc.eval_hash(orc.Env.from_packed(game, ls[0]), m["seed"])
                eng.backup(probs, vals)
        else:
            eng.search()
        rc = eng.root_children()
        k = int(d[name + "__n_children"][ply])
        assert rc["n"][0] == k, (name, ply)
        assert list(rc["actions"][0][:k]) == list(d[name + "__actions"][ply][:k]), (name, ply)
        assert list(rc["visits"][0][:k]) == list(d[name + "__visits"][ply][:k]), (name, ply)
        assert rc["root_visits"][0] == d[name + "__root_N"][ply]
        assert rc["root_q"][:1].view(np.uint32)[0] == d[name + "__root_Q_bits"][ply]
        assert np.array_equal(rc["q"][0][:k].view(np.uint32), d[name + "__child_Q_bits"][ply][:k].astype(np.uint32))
        assert np.array_equal(rc["prior"][0][:k].view(np.uint32), d[name + "__child_P_bits"][ply][:k].astype(np.uint32))
        a, pi = eng.choose_moves(temp=temperature)
        np.testing.assert_allclose(pi[0], d[name + "__pi"][ply][:A], rtol=1e-6, atol=1e-30)
        assert a[0] == d[name + "__action"][ply], (name, ply)
        eng.advance(a)
    _, w, dn = eng.get_roots()
    wref = int(d[name + "__winner"])
    if wref != 99:
        assert dn[0] == 1 and w[0] == wref
    eng.close()


def test_search_golden_fused(E, meta):
    """every recorded reference search (C4 / TTT / NBTTT, flips, tree reuse and fresh trees): fused kernel."""
    d = np.load(os.path.join(G, "search.npz"))
    for name, m in meta["search_cases"].items():
        _run_fixture_case(E, name, m, d, split=False)


def test_search_golden_split_host_eval(E, meta):
    """same vectors through az_select / az_backup with the evaluator on the host (the CNN path's kernels)."""
    d = np.load(os.path.join(G, "search.npz"))
    for name in ("c4_hash_s80_flips_reuse_0", "c4_hash_s120_flips_fresh", "ttt_hash_s100_flips_reuse",
                 "ttt_uniform_s100_reuse", "nbttt_hash_s100_reuse_0"):
        _run_fixture_case(E, name, meta["search_cases"][name], d, split=True)


@pytest.mark.parametrize("game,S,c", [("Connect4", 200, 1.25), ("TTT", 100, 1.0), ("NBTTT", 150, 1.5)])
def test_many_games_vs_oracle(E, game, S, c):
    """256 concurrent games with per-game flip streams, several plies with tree reuse; every game is replayed on
    the oracle: visit counts and Q bits of all root children must match in every slot."""
    rng = np.random.default_rng(11)
    Gn, plies, seed = 256, 4, 4711
    eng = E.SelfPlayEngine(game, n_games=Gn, n_playout=S, c_puct=c, evaluator="hash", eval_seed=seed, flip="stream",
                           reuse_tree=True)
    flips = rng.integers(0, 2, size=(Gn, S * plies)).astype(np.uint8)
    eng.set_flip_stream(flips)
    if game == "NBTTT":
        envs = [orc.Env.nbttt_open(int(b), int(a), int(p)) for b, a, p in rng.integers(0, 9, size=(Gn, 3))]
    else:
        envs = []
        for g in range(Gn):  # random short prefixes so that the games differ
            e = orc.Env(game)
            for _ in range(int(rng.integers(0, 4))):
                e.step(int(rng.choice(e.valid_move())))
            envs.append(e)
    eng.set_roots(np.array([e.packed() for e in envs]))
    trees = []
    for g in range(Gn):
        t = orc.MCTS(game, c_init=c, n_playout=S, eval_kind=orc.EVAL_HASH, eval_seed=seed)
        t.set_flips(flips[g])
        trees.append(t)
    for ply in range(plies):
        eng.search()
        rc = eng.root_children()
        a, pi = eng.choose_moves(temp=1e-3)
        for g in range(Gn):
            if envs[g].done():
                assert rc["n"][g] == 0 or a[g] == -1
                continue
            acts, vis = trees[g].get_action_visits(envs[g])
            k = len(acts)
            assert rc["n"][g] == k
            assert list(rc["actions"][g][:k]) == acts, (g, ply)
            assert list(rc["visits"][g][:k]) == vis, (g, ply)
            ca, cv, cq, cp = trees[g].children(())
            assert np.array_equal(rc["q"][g][:k].view(np.uint32), cq.view(np.uint32))
            opi, oa = orc.move_probs(acts, vis, 1e-3, eng.A)
            assert oa == a[g]
            np.testing.assert_allclose(pi[g], opi, rtol=1e-6, atol=1e-30)
            trees[g].prune_root(oa)
            envs[g].step(oa)
        eng.advance(a)
    st = eng.stats()
    assert st["playouts"] > 0 and st["leaf_evals"] + st["terminal_hits"] == st["playouts"]
    eng.close()


def test_invariants_full_size_ttt(E):
    """BASELINE config 2 shape (4096 TTT games x 100 sims, uniform priors): size-independent invariants.
    fresh tree, S playouts on a non-terminal root: root N = S, sum(child N) = S-1, identical games -> identical trees,
    visits differ by at most 1 across children (uniform priors, value 0), surplus goes to the earliest children."""
    Gn, S = 4096, 100
    eng = E.SelfPlayEngine("TTT", n_games=Gn, n_playout=S, c_puct=1.0, evaluator="uniform", reuse_tree=True)
    eng.search()
    rc = eng.root_children()
    assert (rc["root_visits"] == S).all() and (rc["n"] == 9).all()
    assert (rc["visits"].sum(1) == S - 1).all()
    assert (rc["visits"] == rc["visits"][0]).all()
    assert list(rc["visits"][0]) == [11] * 9  # SURVEY 8c [probed]: TTT uniform/0-value/no-flip move 1
    a, pi = eng.choose_moves(temp=1e-3)
    eng.advance(a)
    eng.search()
    rc = eng.root_children()
    assert list(rc["visits"][0][:8]) == [14, 14, 14, 14, 14, 14, 13, 13]  # SURVEY 8c [probed] move 2
    eng.close()
