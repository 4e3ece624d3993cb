"""K4 + collector parity: whole self-play games on the device vs the oracle's game loop (game.py:35-58),
trajectory packing, the ReplayBuffer-shaped sink, slot refill, and the production (noise / sampling) modes."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from alphazero_amd import engine
    return engine


def _by_game(tr):
    games = {}
    for i in range(len(tr["z"])):
        games.setdefault(int(tr["game_id"][i]), []).append(i)
    return games


@pytest.mark.parametrize("game,S,c,first_n", [("Connect4", 60, 1.25, 10), ("TTT", 40, 1.0, 1), ("NBTTT", 50, 1.5, 12)])
def test_selfplay_trajectories_vs_oracle(E, game, S, c, first_n):
    """deterministic mode (recorded flips, first-max moves, no noise): every 6-tuple ingredient of every game must
    equal the oracle's: packed state before the move, pi (1e-6), mover-relative z, action, done flag, length."""
    rng = np.random.default_rng(3)
    Gn, seed = 32, 2025
    eng = E.SelfPlayEngine(game, n_games=Gn, n_playout=S, c_puct=c, evaluator="hash", eval_seed=seed, flip="stream",
                           reuse_tree=True, move="first_max", rng_seed=5)
    flips = rng.integers(0, 2, size=(Gn, S * 90)).astype(np.uint8)
    eng.set_flip_stream(flips)
    n = eng.selfplay(Gn, temp=1.0, first_n_steps=first_n)
    assert n == Gn
    roots0 = None
    tr = eng.traj_fetch()
    games = _by_game(tr)
    assert sorted(games) == list(range(Gn))
    for g in range(Gn):
        idx = games[g]
        start = orc.Env.from_packed(game, tr["states"][idx[0]])  # NBTTT: device-generated opening
        if game != "NBTTT":
            assert np.array_equal(start.packed(), orc.Env(game).packed())
        t = orc.MCTS(game, c_init=c, n_playout=S, eval_kind=orc.EVAL_HASH, eval_seed=seed)
        t.set_flips(flips[g])
        r = t.selfplay_game(start, temp=1.0, first_n_steps=first_n, reuse_tree=True)
        assert r["n"] == len(idx), (g, r["n"], len(idx))
        assert np.array_equal(tr["states"][idx], r["states"])
        assert np.array_equal(tr["action"][idx], r["actions"].astype(np.int8))
        assert np.array_equal(tr["player"][idx], r["players"].astype(np.int8))
        assert np.array_equal(tr["z"][idx], r["z"])
        np.testing.assert_allclose(tr["pi"][idx], r["pis"], rtol=1e-6, atol=1e-30)
        assert list(tr["done"][idx]) == [0] * (len(idx) - 1) + [1]
    eng.close()


@pytest.mark.parametrize("game,rec_bytes,Gn", [("Connect4", 48, 16), ("TTT", 48, 40), ("NBTTT", 72, 12)])
def test_pack_gather_unpack_replay_tensors(E, game, rec_bytes, Gn):
    """finished plies -> packed records -> (world size 1 gather) -> ReplayBuffer-shaped device tensors; compared with
    the tuples game.py:58 would have produced (state, pi, z, z, next_state, done), for every game's plane format
    (env_cython.pyx:143-149, TTT/utils.py:5-11, NBTTT/utils.py:258-268)."""
    import torch
    from alphazero_amd.collector import ReplaySink, TrajectoryGatherer
    S = 30
    eng = E.SelfPlayEngine(game, n_games=Gn, n_playout=S, evaluator="hash", eval_seed=1, flip="never",
                           move="first_max")
    eng.selfplay(Gn, temp=1.0, first_n_steps=4)
    tr = eng.traj_fetch(clear=False)
    dev = torch.device("cuda", 0)
    recs = TrajectoryGatherer(eng, dev).gather()
    assert recs.shape == (len(tr["z"]), rec_bytes) and eng.traj_count() == (0, 0)
    sink = ReplaySink(game, capacity=len(tr["z"]) + 7, device=dev)
    sink.store_packed(recs)
    n = len(tr["z"])
    assert len(sink) == n and sink.games_seen == Gn and sink.plies_seen == n
    st, ns = sink.state[:n].cpu().numpy(), sink.next_state[:n].cpu().numpy()
    for i in range(n):
        env = orc.Env.from_packed(game, tr["states"][i])
        assert np.array_equal(st[i], env.current_state()[0])
        env.step(int(tr["action"][i]))
        assert np.array_equal(ns[i], env.current_state()[0])
    assert np.array_equal(sink.prob[:n].cpu().numpy(), tr["pi"])
    assert np.array_equal(sink.value[:n, 0].cpu().numpy(), tr["z"])
    assert np.array_equal(sink.winner[:n, 0].cpu().numpy(), tr["z"].astype(np.int32))
    assert np.array_equal(sink.done[:n, 0].cpu().numpy(), tr["done"].astype(bool))
    # ring wrap: storing again overwrites from the pointer on
    sink.store_packed(recs)
    assert sink.total_stored == 2 * n and len(sink) == sink.capacity
    eng.close()


def test_refill_runs_exact_number_of_games(E):
    Gn, target = 16, 50
    eng = E.SelfPlayEngine("TTT", n_games=Gn, n_playout=20, c_puct=1.0, evaluator="hash", eval_seed=9, flip="random",
                           move="sample", rng_seed=77)
    n = eng.selfplay(target, temp=1.0, first_n_steps=3)
    assert n == target
    tr = eng.traj_fetch()
    games = _by_game(tr)
    assert sorted(games) == list(range(target))
    for gid, idx in games.items():
        env = orc.Env("TTT")
        for j, i in enumerate(idx):
            assert np.array_equal(env.packed(), tr["states"][i])
            assert tr["player"][i] == env.turn
            assert int(tr["action"][i]) in env.valid_move()
            assert abs(float(tr["pi"][i].sum()) - 1.0) < 1e-5 and tr["pi"][i][~np.array(env.valid_mask())].sum() == 0
            env.step(int(tr["action"][i]))
            assert env.done() == bool(tr["done"][i])
        w = env.winPlayer()
        want = [0.0 if w == 0 else (1.0 if tr["player"][i] == w else -1.0) for i in idx]
        assert list(tr["z"][idx]) == want
    eng.close()


def test_train_mode_noise_and_random_flips_statistics(E):
    """train() mode (Dirichlet at every expansion, MCTS.py:155-157) is judged statistically: trees stay valid,
    priors of a node still sum to 1, and the noise actually moves them."""
    Gn, S = 512, 120
    base = E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, evaluator="hash", eval_seed=3, flip="never")
    noisy = E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, evaluator="hash", eval_seed=3, flip="random",
                             dirichlet_alpha=0.7, rng_seed=123)
    base.search()
    noisy.search()
    a, b = base.root_children(), noisy.root_children()
    assert (b["visits"].sum(1) == S - 1).all() and (b["root_visits"] == S).all()
    np.testing.assert_allclose(b["prior"].sum(1), 1.0, rtol=1e-5)
    assert (a["prior"] == a["prior"][0]).all()          # same position, no noise -> identical priors
    assert np.abs(b["prior"] - a["prior"]).mean() > 0.01  # noise mixed in with weight 0.25
    # mean of Dirichlet(0.7 x 7) components is 1/7: the average noisy prior stays near 0.75 p + 0.25/7
    # (the empty board is mirror-symmetric, so a flipped root evaluation sees the same position and flip_action
    # is the identity: env_cython.pyx:174-190)
    p = a["prior"][0]
    want = 0.75 * p + 0.25 / 7
    np.testing.assert_allclose(b["prior"].mean(0), want, atol=0.02)
    st = noisy.stats()
    assert st["playouts"] == Gn * S
    base.close()
    noisy.close()


from helpers import RefShapedBuffer as _RefShapedBuffer  # noqa: E402


def test_store_many_equals_per_ply_store(E):
    """collector.store_many (one unpack launch per contiguous piece) must leave a reference-shaped ReplayBuffer in
    exactly the state that Pipeline.data_collector's `for data in play_data: buffer.store(*data)` (Pipeline.py:61-62)
    leaves it in, ring wrap included."""
    import torch
    from alphazero_amd.collector import TrajectoryGatherer, store_many
    Gn, S = 24, 24
    eng = E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, evaluator="hash", eval_seed=5, flip="never",
                           move="first_max")
    eng.selfplay(Gn, temp=1.0, first_n_steps=6)
    tr = eng.traj_fetch(clear=False)
    dev = torch.device("cuda", 0)
    recs = TrajectoryGatherer(eng, dev).gather()
    n = len(tr["z"])
    cap = n - 11                       # smaller than the data: both paths must wrap
    bulk = _RefShapedBuffer(3, cap, 7, 6, 7, dev)
    ply = _RefShapedBuffer(3, cap, 7, 6, 7, dev)
    bulk._ptr = ply._ptr = 5
    first = store_many(bulk, recs, "Connect4")
    assert first == 5
    for i in range(n):                 # the 6-tuples of game.py:58, one store() per ply
        env = orc.Env.from_packed("Connect4", tr["states"][i])
        s = env.current_state()
        env.step(int(tr["action"][i]))
        ply.store(s, tr["pi"][i], float(tr["z"][i]), float(tr["z"][i]), env.current_state(), bool(tr["done"][i]))
    assert bulk._ptr == ply._ptr == (5 + n) % cap
    for name in ("state", "prob", "value", "winner", "next_state", "done"):
        a, b = getattr(bulk, name).cpu().numpy(), getattr(ply, name).cpu().numpy()
        assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f"), name
    eng.close()


@pytest.mark.parametrize("game,dims,Gn", [("Connect4", (3, 7, 6, 7), 48), ("TTT", (3, 9, 3, 3), 80),
                                           ("NBTTT", (28, 9, 3, 3), 16)])
def test_unpack_rows_do_not_depend_on_where_the_ring_pointer_stands(E, game, dims, Gn):
    """az_traj_unpack_dev peels the plies before the first 128-byte line boundary and the ragged tail off to a
    narrower kernel and streams the middle in 32-ply chunks: whatever row the destination starts at (every residue
    mod 16 moves the boundary) and whatever the count, the rows written are the rows of the pinned case above and
    nothing outside [lo, lo + n) is touched."""
    import torch
    from alphazero_amd.collector import TrajectoryGatherer, store_many
    eng = E.SelfPlayEngine(game, n_games=Gn, n_playout=16, evaluator="hash", eval_seed=11, flip="never",
                           move="first_max")
    eng.selfplay(Gn, temp=1.0, first_n_steps=8)
    dev = torch.device("cuda", 0)
    recs = TrajectoryGatherer(eng, dev).gather()
    eng.close()
    n = int(recs.shape[0])
    assert n > 300
    C_, A_, H_, W_ = dims
    ref = _RefShapedBuffer(C_, n, A_, H_, W_, dev)
    store_many(ref, recs, game)
    names = ("state", "prob", "value", "winner", "next_state", "done")
    for lo, m in [(0, n), (1, n), (3, n - 1), (7, 100), (8, 31), (15, 33), (16, 64), (17, 65), (29, n - 5), (2, 1),
                  (2, 0)]:
        buf = _RefShapedBuffer(C_, n + 64, A_, H_, W_, dev)
        buf.winner.fill_(-7)
        buf.done.fill_(True)
        buf._ptr = lo
        store_many(buf, recs[:m], game)
        assert buf._ptr == lo + m
        for name in names:
            got, want = getattr(buf, name), getattr(ref, name)
            assert torch.equal(got[lo:lo + m], want[:m]), (lo, m, name)     # written rows hold no NaN
            rest = torch.cat([got[:lo], got[lo + m:]])
            if got.dtype == torch.float32:
                assert torch.isnan(rest).all(), (lo, m, name)
            elif got.dtype == torch.int32:
                assert (rest == -7).all(), (lo, m, name)
            else:
                assert rest.all(), (lo, m, name)
