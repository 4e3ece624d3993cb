"""Pins the CPU oracle (oracle/az_oracle*.c) against golden vectors recorded from the reference itself
(tests/golden/make_fixtures.py).  CPU-only; no GPU, no /root/reference needed."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GAMES = {"Connect4": orc.C4, "TTT": orc.TTT, "NBTTT": orc.NBTTT}


@pytest.fixture(scope="module")
def traces():
    return np.load(os.path.join(G, "env_traces.npz"))


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(G, "fixtures.json")) as f:
        return json.load(f)


def ref_cells_to_oracle(game, cells):
    return np.asarray(cells, dtype=np.int8)


@pytest.mark.parametrize("game", ["Connect4", "TTT", "NBTTT"])
def test_env_traces_bit_exact(traces, game):
    """env transitions, legal masks, winners, done flags and state planes: bit-exact."""
    starts = traces[game + "_game_start"]
    action = traces[game + "_action"]
    n_checked = 0
    for g in range(len(starts) - 1):
        lo, hi = starts[g], starts[g + 1]
        if game == "NBTTT":
            env = orc.Env.nbttt_open(*[int(x) for x in traces["NBTTT_opening"][g]])
        else:
            env = orc.Env(game)
        for i in range(lo, hi):
            if action[i] >= 0:
                assert action[i] in env.valid_move()
                env.step(int(action[i]))
            assert np.array_equal(env.cells(), traces[game + "_cells"][i]), (g, i)
            assert env.turn == traces[game + "_turn"][i]
            assert np.array_equal(np.array(env.valid_mask(), dtype=np.uint8), traces[game + "_mask"][i])
            assert env.winPlayer() == traces[game + "_winner"][i]
            assert env.done() == bool(traces[game + "_done"][i])
            assert np.array_equal(env.current_state().ravel(), traces[game + "_state"][i])
            if game == "NBTTT":
                assert env.curr == traces["NBTTT_curr"][i]
            # pack / unpack round trip of the public bitboard format
            rt = orc.Env.from_packed(game, env.packed())
            assert np.array_equal(rt.cells(), env.cells()) and rt.turn == env.turn and rt.curr == env.curr
            n_checked += 1
    assert n_checked > 1000


def test_c4_full_column_noop(traces):
    env = orc.Env.from_cells("Connect4", traces["c4_fullcol_before_cells"])
    env.turn = int(traces["c4_fullcol_before_turn"])
    assert env.place(3) is False
    env.step(3)
    assert np.array_equal(env.cells(), traces["c4_fullcol_after_cells"])
    assert env.turn == int(traces["c4_fullcol_after_turn"]) == int(traces["c4_fullcol_before_turn"])


def test_c4_random_boards(traces):
    cells = traces["c4_rand_cells"]
    both = 0
    for i in range(len(cells)):
        env = orc.Env.from_cells("Connect4", cells[i])
        assert env.turn == traces["c4_rand_turn"][i]
        assert env.winPlayer() == traces["c4_rand_winner"][i]
        assert env.check_full() == bool(traces["c4_rand_full"][i])
        assert [env.flip_action(a) for a in range(7)] == list(traces["c4_rand_flip_action"][i])
        assert np.array_equal(np.array(env.valid_mask(), dtype=np.uint8), traces["c4_rand_mask"][i])
        both += 1
    assert both == 600


def test_ttt_nbttt_random_boards(traces):
    for i, c in enumerate(traces["ttt_rand_cells"]):
        env = orc.Env("TTT")
        for k in range(9):
            env.e.cell[k] = int(c[k])
        assert env.winPlayer() == traces["ttt_rand_winner"][i]
    assert [orc.Env("TTT").flip_action(a) for a in range(9)] == list(traces["ttt_flip_action"])
    for i, c in enumerate(traces["nbttt_rand_cells"]):
        env = orc.Env("NBTTT")
        for k in range(100):
            env.e.cell[k] = int(c[k])
        assert env.winPlayer() == traces["nbttt_rand_winner"][i]


def test_puct_bit_exact():
    """TreeNode.PUCT float32 rounding sequence (MCTS.py:47-55): bit patterns."""
    d = np.load(os.path.join(G, "puct.npz"))
    n = len(d["Np"])
    got = np.array([orc.puct(float(d["c_init"][i]), 500, int(d["Np"][i]), int(d["n"][i]),
                             float(d["Q"][i]) if d["n"][i] > 0 else 0.0, float(d["P"][i])) for i in range(n)],
                   dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), d["score_bits"])


def test_move_probs():
    """player.py:101-111: pi within 1e-6 relative (float64 exp/log -> float32), choices exact."""
    d = np.load(os.path.join(G, "moves.npz"))
    for i in range(len(d["n"])):
        k = int(d["n"][i])
        A = int(d["n_actions"][i])
        pi, chosen = orc.move_probs(d["actions"][i][:k], d["visits"][i][:k], float(d["temp"][i]), A)
        np.testing.assert_allclose(pi, d["pi"][i][:A], rtol=1e-6, atol=1e-30)
        assert chosen == d["chosen"][i]


def _run_case(name, m, S):
    game = m["game"]
    d = np.load(os.path.join(G, "search.npz"))
    kind = orc.EVAL_UNIFORM if m["kind"] == "uniform" else orc.EVAL_HASH
    t = orc.MCTS(game, c_init=m["c_puct"], n_playout=m["S"], eval_kind=kind, eval_seed=m["seed"])
    if m["has_flips"]:
        t.set_flips(d[name + "__flips"])
    if game == "NBTTT":
        env = orc.Env.nbttt_open(*m["opening"])
    else:
        env = orc.Env(game)
    A = env.n_actions
    plies = len(d[name + "__action"])
    for ply in range(plies):
        assert not env.done()
        first_n, temp = m["first_n"], m["temp"]
        temperature = (1e-3 if ply >= first_n else temp) if first_n is not None else temp
        acts, vis = t.get_action_visits(env)
        k = int(d[name + "__n_children"][ply])
        assert acts == list(d[name + "__actions"][ply][:k]), (name, ply)
        assert vis == list(d[name + "__visits"][ply][:k]), (name, ply)
        assert t.root_visits == d[name + "__root_N"][ply]
        assert t.root_q.view(np.uint32) == d[name + "__root_Q_bits"][ply]
        ca, cv, cq, cp = t.children(())
        assert np.array_equal(cq.view(np.uint32), d[name + "__child_Q_bits"][ply][:k].astype(np.uint32))
        assert np.array_equal(cp.view(np.uint32), d[name + "__child_P_bits"][ply][:k].astype(np.uint32))
        pi, chosen = orc.move_probs(acts, vis, temperature, A)
        np.testing.assert_allclose(pi, d[name + "__pi"][ply][:A], rtol=1e-6, atol=1e-30)
        assert chosen == d[name + "__action"][ply]
        if m["reuse"]:
            t.prune_root(chosen)
        else:
            t.reset()
        env.step(chosen)
    w = int(d[name + "__winner"])
    if w != 99:
        assert env.done() and env.winPlayer() == w


def test_search_visits_bit_exact(meta):
    """MCTS_AZ.get_action_visits with injected evaluators: actions (insertion order), visit counts,
    root/child Q and prior BIT PATTERNS, per ply, over whole games, all three games."""
    for name, m in meta["search_cases"].items():
        _run_case(name, m, m["S"])


def test_selfplay_trajectory():
    """game.py:35-58 6-tuples of one self-play game (state, pi, z, z, next_state, done)."""
    d = np.load(os.path.join(G, "trajectory.npz"))
    t = orc.MCTS("Connect4", c_init=1.25, n_playout=60, eval_kind=orc.EVAL_HASH, eval_seed=2024)
    t.set_flips(d["flips"])
    r = t.selfplay_game(orc.Env("Connect4"), temp=1.0, first_n_steps=10, reuse_tree=True)
    n = r["n"]
    assert n == len(d["z"]) and r["winner"] == int(d["winner"])
    np.testing.assert_array_equal(r["z"], d["z"])
    np.testing.assert_array_equal(d["z"], d["z2"])
    np.testing.assert_allclose(r["pis"], d["pi"], rtol=1e-6, atol=1e-30)
    assert list(d["done"]) == [False] * (n - 1) + [True]
    for i in range(n):
        env = orc.Env.from_packed("Connect4", r["states"][i])
        assert np.array_equal(env.current_state()[0], d["state"][i])
        env.step(int(r["actions"][i]))
        assert np.array_equal(env.current_state()[0], d["next_state"][i])


@pytest.fixture(scope="module")
def blob():
    b = np.fromfile(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    assert b.size == orc.CNN_BLOB_FLOATS
    return b


def test_weights_blob_checksum(meta, blob):
    import hashlib
    assert hashlib.sha256(blob.tobytes()).hexdigest() == meta["weights"]["sha256"]


def test_cnn_forward_within_1e4(blob):
    """CNN.forward / predict (Network.py:60-85): log-probs and value within 1e-4 relative of torch fp32."""
    d = np.load(os.path.join(G, "cnn.npz"))
    net = orc.CNN(blob)
    states = []
    for c, t in zip(d["cells"], d["turn"]):
        env = orc.Env.from_cells("Connect4", c)
        assert env.turn == t
        states.append(env.current_state()[0])
    states = np.array(states)
    logp, logv = net.forward(states)
    probs, value = net.predict(states)
    np.testing.assert_allclose(logp, d["logp"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logv, d["logv"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(probs, d["probs"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(value, d["value"], rtol=1e-4, atol=1e-5)


def test_realnet_search_moves(blob):
    """BASELINE config 1 head (real CNN, S=500, eval, fresh tree per ply, recorded flips): the oracle's own
    fp32 forward differs from torch's in the last bits, so visits are compared as move agreement + L1."""
    d = np.load(os.path.join(G, "realnet_search.npz"))
    net = orc.CNN(blob)
    S = int(d["S"])
    plies = 3  # bounded: ~1.5k batch-1 forwards
    t = orc.MCTS("Connect4", c_init=1.25, n_playout=S, eval_kind=orc.EVAL_CNN, net=net)
    t.set_flips(d["flips"])
    env = orc.Env("Connect4")
    for ply in range(plies):
        acts, vis = t.get_action_visits(env)
        ref_acts = [a for a in d["actions"][ply] if a >= 0]
        assert acts == ref_acts
        ref_vis = d["visits"][ply][:len(acts)]
        chosen = acts[int(np.argmax(vis))]
        assert chosen == d["chosen"][ply]
        assert np.abs(np.array(vis) - ref_vis).sum() <= 0.05 * S
        t.reset()
        env.step(chosen)


# ---------------------------------------------------------------------------- pure MCTS (UCT + rollout, the Elo opponent)
def _pure_start_env(m):
    if m["game"] == "NBTTT":
        return orc.Env.nbttt_open(*m["opening"])
    return orc.Env(m["game"])


def test_uct_bit_exact_vs_reference():
    """TreeNode.UCT (MCTS.py:57-63): float64 bit patterns recorded from the reference."""
    d = np.load(os.path.join(G, "pure_mcts.npz"))
    got = np.array([orc.uct(c, 500, Np, n, Q) for c, Np, n, Q in zip(d["uct_c"], d["uct_Np"], d["uct_n"], d["uct_Q"])])
    assert np.array_equal(got.view(np.uint64), d["uct_bits"])


def test_pure_mcts_vs_reference_games():
    """MCTS.get_action (MCTS.py:115-128) with utils.policy_value_fn, ply by ply: root children in insertion order,
    visit counts, float64 Q bit patterns, root statistics and the returned action -- all exact."""
    d = np.load(os.path.join(G, "pure_mcts.npz"))
    with open(os.path.join(G, "pure_fixtures.json")) as f:
        meta = json.load(f)["cases"]
    for name, m in meta.items():
        t = orc.PureMCTS(m["game"], m["c_puct"], m["S"], seed=m["seed"])
        t.set_flips(d[name + "__flips"] if m["has_flips"] else None)
        env = _pure_start_env(m)
        plies = min(m["plies"], 14)   # the first plies carry the big trees; the GPU test replays every ply
        for ply in range(plies):
            best, acts, vis, q = t.get_action(env)
            k = int(d[name + "__n_children"][ply])
            assert acts == list(d[name + "__actions"][ply][:k]), (name, ply)
            assert vis == list(d[name + "__visits"][ply][:k]), (name, ply)
            assert np.array_equal(q.view(np.uint64), d[name + "__q_bits"][ply][:k].astype(np.uint64)), (name, ply)
            assert t.root_visits == d[name + "__root_N"][ply]
            assert np.float64(t.root_q).view(np.uint64) == d[name + "__root_Q_bits"][ply]
            assert best == d[name + "__action"][ply], (name, ply)
            t.prune_root(best if m["reuse"] else -1)
            env.step(best)
        w = int(d[name + "__winner"])
        if w != 99 and plies == m["plies"]:
            assert env.done() and env.winPlayer() == w
