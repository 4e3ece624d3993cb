"""Pure Monte-Carlo tree search on the device (UCT selection in float64 + rollouts; SURVEY 8f-3, the Elo arena's opponent
MCTSPlayer): HIP through the C ABI vs vectors recorded from the reference's MCTS class (tests/golden/pure_mcts.npz, made
by make_fixtures_pure.py) and vs the CPU oracle.  Visit counts, child order, chosen action: exact.  Q: the float64
values rounded to float32 (the ABI reports Q as float)."""
import json
import os
import sys

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
DROPIN = os.path.join(ROOT, "alphazero_amd", "dropin")


@pytest.fixture(scope="module")
def E():
    from alphazero_amd import engine
    return engine


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(G, "pure_fixtures.json")) as f:
        return json.load(f)["cases"], np.load(os.path.join(G, "pure_mcts.npz"))


def _start_env(m):
    return orc.Env.nbttt_open(*m["opening"]) if m["game"] == "NBTTT" else orc.Env(m["game"])


def test_pure_search_vs_reference_games(E, cases):
    meta, d = cases
    for name, m in meta.items():
        eng = E.SelfPlayEngine(m["game"], n_games=1, n_playout=m["S"], c_puct=m["c_puct"], evaluator="rollout_hash",
                               eval_seed=m["seed"], flip="stream" if m["has_flips"] else "never", reuse_tree=m["reuse"])
        if m["has_flips"]:
            eng.set_flip_stream(d[name + "__flips"].reshape(1, -1))
        eng.set_roots(_start_env(m).packed().reshape(1, -1))
        for ply in range(m["plies"]):
            eng.search()
            rc = eng.root_children()
            k = int(d[name + "__n_children"][ply])
            assert rc["n"][0] == k, (name, ply)
            assert list(rc["actions"][0][:k]) == list(d[name + "__actions"][ply][:k]), (name, ply)
            assert list(rc["visits"][0][:k]) == list(d[name + "__visits"][ply][:k]), (name, ply)
            want_q = d[name + "__q_bits"][ply][:k].astype(np.uint64).view(np.float64).astype(np.float32)
            assert np.array_equal(rc["q"][0][:k], want_q), (name, ply)
            assert rc["root_visits"][0] == d[name + "__root_N"][ply]
            a, _ = eng.choose_moves(temp=0.0)       # first maximum of the visit counts (MCTS.py:128)
            assert a[0] == d[name + "__action"][ply], (name, ply)
            eng.advance(a)
        w = int(d[name + "__winner"])
        _, win, done = eng.get_roots()
        if w != 99:
            assert done[0] == 1 and win[0] == w
        eng.close()


@pytest.mark.parametrize("game,S,c", [("Connect4", 300, 1.0), ("TTT", 120, 4.0), ("NBTTT", 200, 1.5)])
def test_pure_many_games_vs_oracle(E, game, S, c):
    """192 concurrent games from different positions with per-game flip streams, 3 plies with tree reuse, every slot
    replayed on the oracle."""
    rng = np.random.default_rng(17)
    Gn, plies, seed = 192, 3, 909
    eng = E.SelfPlayEngine(game, n_games=Gn, n_playout=S, c_puct=c, evaluator="rollout_hash", eval_seed=seed,
                           flip="stream", reuse_tree=True)
    flips = rng.integers(0, 2, size=(Gn, S * plies)).astype(np.uint8)
    eng.set_flip_stream(flips)
    if game == "NBTTT":
        envs = [orc.Env.nbttt_open(int(b), int(a), int(p)) for b, a, p in rng.integers(0, 9, size=(Gn, 3))]
    else:
        envs = []
        for g in range(Gn):
            e = orc.Env(game)
            for _ in range(int(rng.integers(0, 5))):
                if not e.done():
                    e.step(int(rng.choice(e.valid_move())))
            envs.append(e)
    eng.set_roots(np.array([e.packed() for e in envs]))
    trees = []
    for g in range(Gn):
        t = orc.PureMCTS(game, c, S, seed=seed)
        t.set_flips(flips[g])
        trees.append(t)
    for ply in range(plies):
        eng.search()
        rc = eng.root_children()
        a, _ = eng.choose_moves(temp=0.0)
        for g in range(Gn):
            if envs[g].done():
                continue
            best, acts, vis, q = trees[g].get_action(envs[g])
            k = len(acts)
            assert rc["n"][g] == k and list(rc["actions"][g][:k]) == acts, (g, ply)
            assert list(rc["visits"][g][:k]) == vis, (g, ply)
            assert np.array_equal(rc["q"][g][:k], q.astype(np.float32)), (g, ply)
            assert a[g] == best, (g, ply)
            trees[g].prune_root(best)
            envs[g].step(best)
        eng.advance(a)
    st = eng.stats()
    assert st["playouts"] > 0 and st["leaf_evals"] + st["terminal_hits"] == st["playouts"] and st["rollout_plies"] > 0
    eng.close()


def test_random_rollouts_statistics(E):
    """AZ_EVAL_ROLLOUT (device RNG, the production mode): statistical parity.  From a position where X wins at once in
    column 3, UCT must put most visits on that move and see it as lost for the opponent (child Q = -1 exactly)."""
    e = orc.Env("Connect4")
    for a in (3, 0, 3, 0, 3, 1):      # X has three in column 3, O scattered; X to move
        e.step(a)
    Gn = 64
    eng = E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=400, c_puct=1.0, evaluator="rollout", flip="random",
                           rng_seed=5)
    eng.set_roots(np.tile(e.packed(), (Gn, 1)))
    eng.search()
    rc = eng.root_children()
    a, _ = eng.choose_moves(temp=0.0)
    assert (a == 3).all()
    for g in range(Gn):
        i = list(rc["actions"][g][:rc["n"][g]]).index(3)
        assert rc["q"][g][i] == -1.0 and rc["visits"][g][i] == rc["visits"][g].max() and rc["visits"][g][i] > 100
    assert len({tuple(v) for v in rc["visits"]}) > 1     # different RNG streams per game
    st = eng.stats()
    assert st["playouts"] == Gn * 400 and st["rollout_plies"] > 0
    eng.close()


def test_dropin_mcts_player_vs_reference(cases):
    """player.MCTSPlayer / MCTS.get_action of the drop-in driven like the reference (player.py:71-82, game.py:17-33)
    with the harness rules of make_fixtures_pure.py."""
    from test_gpu_dropin import RngPatch
    meta, d = cases
    sys.path.insert(0, DROPIN)
    try:
        import environments
        import player
        from utils import RolloutPolicy
        for name in ("ttt_s100_c4_flips_fresh", "c4_s200_c1_flips_fresh"):
            m = meta[name]
            p = player.MCTSPlayer(m["c_puct"], m["S"], policy=RolloutPolicy(hash_seed=m["seed"]))
            env = environments.load(m["game"]).Env()
            plies = min(m["plies"], 12)
            with RngPatch([int(x) for x in d[name + "__flips"]]):
                for ply in range(plies):
                    action, child, *_ = p.get_action(env)
                    p.reset_player()
                    assert action == d[name + "__action"][ply], (name, ply)
                    k = int(d[name + "__n_children"][ply])
                    i = list(d[name + "__actions"][ply][:k]).index(action)
                    assert child.n_visits == d[name + "__visits"][ply][i]
                    env.step(action)
    finally:
        sys.path.remove(DROPIN)


def test_arena_alphazero_beats_pure_mcts(E):
    """Pipeline.update_elo's pairing (Pipeline.py:125-135) as a batch: the trained network with 64 playouts against
    pure MCTS with 200 rollouts, 48 games each way.  The network player must win clearly (statistical)."""
    from alphazero_amd.arena import play_arena
    blob = np.fromfile(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    Gn = 48
    az = E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=64, c_puct=1.25, evaluator="cnn", flip="random",
                          reuse_tree=False, move="sample", rng_seed=1)
    az.load_net(blob)
    pure = E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=200, c_puct=1.0, evaluator="rollout", flip="random",
                            reuse_tree=False, move="sample", rng_seed=2)
    w1, p1 = play_arena(az, pure)       # AlphaZero plays X
    w2, p2 = play_arena(pure, az)       # AlphaZero plays O
    az_wins = int((w1 == 1).sum() + (w2 == -1).sum())
    pure_wins = int((w1 == -1).sum() + (w2 == 1).sum())
    print("arena: AlphaZero %d, pure MCTS %d, draws %d of %d games; mean plies %.1f"
          % (az_wins, pure_wins, 2 * Gn - az_wins - pure_wins, 2 * Gn, float(np.concatenate([p1, p2]).mean())))
    assert (p1 >= 7).all() and (p2 >= 7).all()
    assert az_wins > 3 * pure_wins
    az.close()
    pure.close()


def test_network_player_arena(E):
    """Pipeline.select_best_player's pairing (Pipeline.py:137-163) as a batch: NetworkPlayer(net, False) against a
    NetworkPlayer with damaged weights.  The intact net must score well above the threshold the reference uses (0.65)."""
    from alphazero_amd.arena import NetworkSide, play_sides, win_rate
    blob = np.fromfile(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    rng = np.random.default_rng(0)
    bad = blob.copy()
    bad[-(7 * 126 + 7 + 128 + 4 + 3 * 42 + 3):] = rng.normal(0, 0.05, size=7 * 126 + 7 + 128 + 4 + 3 * 42 + 3)  # heads
    Gn = 128
    cur = E.SelfPlayEngine("Connect4", n_games=1, n_playout=1, evaluator="cnn")
    cur.load_net(blob)
    old = E.SelfPlayEngine("Connect4", n_games=1, n_playout=1, evaluator="cnn")
    old.load_net(bad)
    a = NetworkSide(cur, deterministic=False, seed=1)
    b = NetworkSide(old, deterministic=False, seed=2)
    wx, px = play_sides("Connect4", Gn, 2, a, b)
    wo, po = play_sides("Connect4", Gn, 2, b, a)
    wr = win_rate(wx, wo)
    print("network arena: win rate of the intact net %.3f over %d games, mean plies %.1f" % (wr, 2 * Gn, np.concatenate([px, po]).mean()))
    assert (px >= 7).all() and wr > 0.65
    # deterministic players: every game of the batch is the same game
    wd, pd = play_sides("Connect4", 8, 2, NetworkSide(cur, True), NetworkSide(cur, True))
    assert (wd == wd[0]).all() and (pd == pd[0]).all()
    cur.close()
    old.close()
