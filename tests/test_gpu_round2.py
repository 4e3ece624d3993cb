"""Round-2 parity and size coverage on the GPU, all through the C ABI:
  * K4 (az_choose_moves) against every vector recorded from AlphaZeroPlayer.get_action, temperature 0 included;
  * weight hot reload on a live engine and a live MCTS_AZ (Pipeline.py:87 -> the next data_collector);
  * collect_self_play rows replayed on the oracle's environment; successive collections differ;
  * the trajectory buffer when it overflows: only committed rows come back;
  * out-of-range actions / current-board nibbles through the batch K1 API;
  * BASELINE configs[3] per-GPU shape (16 384 Connect4 slots x 500 playouts, real net) with pack -> 1-rank RCCL gather ->
    store_many, and configs[4] shape (NBTTT, 800 playouts, tens of GB of node pool);
  * bench.py's own N-rank launcher on one GPU.
"""
import json
import os
import subprocess
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
def blob():
    return np.fromfile(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")


@pytest.fixture(scope="module")
def blob_b():
    return np.fromfile(os.path.join(G, "weights", "AZ2_Connect4_CNN_best.f32"), dtype="<f4")


@pytest.fixture(scope="module")
def dropin():
    sys.path.insert(0, DROPIN)
    import environments
    import game
    import MCTS
    import player
    yield dict(load=environments.load, game=game, MCTS=MCTS, player=player)
    sys.path.remove(DROPIN)


# ------------------------------------------------------------------------------------------------ K4, all temperatures
@pytest.mark.parametrize("game,A", [("Connect4", 7), ("TTT", 9)])
def test_choose_moves_vs_recorded_get_action(E, game, A):
    """tests/golden/moves.npz: (actions, visits, T) -> (pi, move) recorded from the reference's
    AlphaZeroPlayer.get_action (player.py:101-111) for T in {0, 1e-3, 1}, ties included.  The device K4 must return pi
    within 1e-6 relative (float64 log/exp -> float32) -- exactly uniform over the maxima at T = 0 -- and the very move."""
    d = np.load(os.path.join(G, "moves.npz"))
    rows = np.flatnonzero(d["n_actions"] == A)
    n = len(rows)
    eng = E.SelfPlayEngine(game, n_games=n, n_playout=8, evaluator="uniform", move="first_max")
    acts = np.where(d["actions"][rows][:, :A] < 0, 0, d["actions"][rows][:, :A])
    eng.set_root_children(acts, d["visits"][rows][:, :A], d["n"][rows])
    a, pi = eng.choose_moves(temps=d["temp"][rows])
    assert np.array_equal(a, d["chosen"][rows])
    np.testing.assert_allclose(pi, d["pi"][rows][:, :A], rtol=1e-6, atol=1e-30)
    t0 = d["temp"][rows] == 0
    assert t0.sum() > 50
    for i in np.flatnonzero(t0):   # T = 0: the distribution handed back is uniform over the arg-max set, bit for bit
        k = int(d["n"][rows[i]])
        v = d["visits"][rows[i]][:k]
        want = np.zeros(A, np.float32)
        want[d["actions"][rows[i]][:k][v == v.max()]] = np.float32(1.0 / (v == v.max()).sum())
        assert np.array_equal(pi[i], want), i
    # the root's children come back as they were set (insertion order)
    rc = eng.root_children()
    assert np.array_equal(rc["n"], d["n"][rows])
    for i in range(0, n, 17):
        k = int(rc["n"][i])
        assert list(rc["actions"][i][:k]) == list(d["actions"][rows[i]][:k])
        assert list(rc["visits"][i][:k]) == list(d["visits"][rows[i]][:k])
    eng.close()


def test_choose_moves_sampling_at_temp_zero_is_uniform_over_maxima(E):
    """production move mode at T = 0 (player.py:106-108: np.random.choice over the arg-max set): only maxima are ever
    chosen, and each of them about equally often"""
    n = 4096
    eng = E.SelfPlayEngine("Connect4", n_games=n, n_playout=8, evaluator="uniform", move="sample", rng_seed=11)
    acts = np.tile(np.array([6, 0, 3, 2, 5, 1, 4]), (n, 1))
    vis = np.tile(np.array([9, 40, 3, 40, 1, 40, 7]), (n, 1))
    eng.set_root_children(acts, vis, np.full(n, 7))
    a, pi = eng.choose_moves(temp=0.0)
    assert set(np.unique(a)) == {0, 2, 1}
    cnt = np.array([(a == x).sum() for x in (0, 2, 1)])
    assert (np.abs(cnt - n / 3) < 5 * np.sqrt(n * 2 / 9)).all(), cnt
    assert np.array_equal(pi[0], np.array([1 / 3, 1 / 3, 1 / 3, 0, 0, 0, 0], dtype=np.float32))
    eng.close()


# ------------------------------------------------------------------------------------------------ hot reload
def test_hot_reload_on_live_engine_and_live_mcts(E, dropin, blob, blob_b):
    """Pipeline.py:87 saves the net after every update and the next data_collector must play with it.  Loading checkpoint
    B over A on a LIVE engine (after searches with A) and on a live MCTS_AZ / AlphaZeroPlayer must leave exactly the
    engine a fresh load of B gives: network outputs, root priors and visit counts bit for bit."""
    rng = np.random.default_rng(5)
    roots = []
    for g in range(32):
        e = orc.Env("Connect4")
        for _ in range(int(rng.integers(0, 12))):
            if not e.done():
                e.step(int(rng.choice(e.valid_move())))
        roots.append(e.packed() if not e.done() else orc.Env("Connect4").packed())
    roots = np.array(roots)

    def searched(eng):
        eng.set_roots(roots)
        eng.search()
        rc = eng.root_children()
        return rc["actions"].copy(), rc["visits"].copy(), rc["prior"].view(np.uint32).copy(), rc["q"].view(np.uint32).copy()

    for kernel in ("wino", "split"):
        live = E.SelfPlayEngine("Connect4", n_games=32, n_playout=64, c_puct=1.25, evaluator="cnn", flip="never")
        live.load_net(blob, kernel=kernel)
        with_a = searched(live)
        fa = live.net_forward(roots)
        live.load_net(blob_b)                       # hot reload; the kernel choice is kept
        assert live.k3_kernel == kernel
        fresh = E.SelfPlayEngine("Connect4", n_games=32, n_playout=64, c_puct=1.25, evaluator="cnn", flip="never")
        fresh.load_net(blob_b, kernel=kernel)
        fb_live, fb_fresh = live.net_forward(roots), fresh.net_forward(roots)
        for x, y in zip(fb_live, fb_fresh):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        assert not np.array_equal(fa[2], fb_live[2])      # and B is not A
        got, want = searched(live), searched(fresh)
        for x, y in zip(got, want):
            assert np.array_equal(x, y)
        assert not np.array_equal(with_a[2], got[2])
        live.close()
        fresh.close()

    # the drop-in: CNN.sync_from / load_state_dict reaches the engines of players that hold the net
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import blob_to_state_dict
    module = dropin["load"]("Connect4")
    net = module.CNN(0, device="cuda")
    net.load(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"))
    pl = dropin["player"].AlphaZeroPlayer(net, c_puct=1.25, n_playout=48, alpha=0.7, is_selfplay=0)
    pl.eval()
    env = module.Env()
    np.random.seed(3)
    pl.get_action(env)                                   # the player's engine now exists and holds checkpoint A
    p_a, _ = net.predict(env.current_state())
    net.load_state_dict(blob_to_state_dict(blob_b))      # what sync_from(module) does after policy_update
    np.random.seed(4)
    acts, vis = pl.mcts.get_action_visits(env)
    p_b, v_b = net.predict(env.current_state())
    net2 = module.CNN(0, device="cuda")
    net2.load(os.path.join(G, "weights", "AZ2_Connect4_CNN_best.f32"))
    pl2 = dropin["player"].AlphaZeroPlayer(net2, c_puct=1.25, n_playout=48, alpha=0.7, is_selfplay=0)
    pl2.eval()
    np.random.seed(4)
    acts2, vis2 = pl2.mcts.get_action_visits(module.Env())
    p2, v2 = net2.predict(env.current_state())
    assert np.array_equal(p_b, p2) and np.array_equal(v_b, v2) and not np.array_equal(p_a, p_b)
    assert acts == acts2 and vis == vis2
    pr, pr2 = pl.mcts.root, pl2.mcts.root
    assert [c.prior for c in pr.children.values()] == [c.prior for c in pr2.children.values()]


# ------------------------------------------------------------------------------------------------ collect_self_play
def test_collect_self_play_rows_replay_on_the_oracle_env(dropin):
    """Game.collect_self_play (= Pipeline.data_collector, Pipeline.py:53-63) at the value level: every stored game is
    replayed ply by ply on the oracle's environment (the reference's own board representation): `state` must be the
    planes of the position, the move is the one cell that `next_state` adds, `next_state` the planes after it, z the
    outcome from each mover's perspective (game.py:52-55), winner = z, done only on the last ply, every game starts from
    the empty board and ends exactly when the oracle env says the game is over.  Two calls in a row must not replay the
    same random draws (the reference draws from the unseeded global np.random)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import RefShapedBuffer
    module = dropin["load"]("Connect4")
    net = module.CNN(0, device="cuda")
    net.load(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"))
    sp = dropin["player"].AlphaZeroPlayer(net, c_puct=1.25, n_playout=32, alpha=0.7, is_selfplay=1)
    sp.train()
    game = dropin["game"].Game(module.Env())
    bufs = []
    for call in range(2):
        buf = RefShapedBuffer(3, 4096, 7, 6, 7, "cuda")
        lens = game.collect_self_play(sp, buf, 48, temp=1.0, first_n_steps=10)
        n = sum(lens)
        assert len(lens) == 48 and buf._ptr == n
        st, nx = buf.state[:n].cpu().numpy(), buf.next_state[:n].cpu().numpy()
        pr, z = buf.prob[:n].cpu().numpy(), buf.value[:n, 0].cpu().numpy()
        wn, dn = buf.winner[:n, 0].cpu().numpy(), buf.done[:n, 0].cpu().numpy()
        lo = 0
        for ln in lens:
            env = orc.Env("Connect4")
            movers = []
            for i in range(lo, lo + ln):
                assert not env.done()
                assert np.array_equal(st[i], env.current_state()[0]), (call, i)
                added = (nx[i][0] + nx[i][1]) - (st[i][0] + st[i][1])
                cols = np.flatnonzero(added.sum(0))
                assert added.sum() == 1 and len(cols) == 1
                a = int(cols[0])
                assert a in env.valid_move() and pr[i][a] > 0            # the move had search probability
                assert abs(pr[i].sum() - 1) < 1e-5 and (pr[i][[c for c in range(7) if c not in env.valid_move()]] == 0).all()
                movers.append(env.turn)
                env.step(a)
                assert np.array_equal(nx[i], env.current_state()[0]), (call, i)
                assert bool(dn[i]) == (i == lo + ln - 1) == env.done()
            w = env.winPlayer()
            want = np.array([0.0 if w == 0 else (1.0 if m == w else -1.0) for m in movers], dtype=np.float32)
            assert np.array_equal(z[lo:lo + ln], want) and np.array_equal(wn[lo:lo + ln], want.astype(np.int32))
            lo += ln
        bufs.append((n, st.copy(), pr.copy()))
    (n1, s1, p1), (n2, s2, p2) = bufs
    m = min(n1, n2)
    assert n1 != n2 or not (np.array_equal(s1[:m], s2[:m]) and np.array_equal(p1[:m], p2[:m]))
    # and a fixed seed does replay
    b1, b2 = RefShapedBuffer(3, 2048, 7, 6, 7, "cuda"), RefShapedBuffer(3, 2048, 7, 6, 7, "cuda")
    l1 = game.collect_self_play(sp, b1, 16, temp=1.0, first_n_steps=10, rng_seed=42)
    l2 = game.collect_self_play(sp, b2, 16, temp=1.0, first_n_steps=10, rng_seed=42)
    assert l1 == l2 and torch.equal(b1.prob[:sum(l1)], b2.prob[:sum(l2)])


# ------------------------------------------------------------------------------------------------ trajectory overflow
def test_trajectory_overflow_returns_only_committed_games(E):
    """the committed-trajectory arrays hold 2 x slots x max plies rows; a caller that never fetches overflows them.  The
    overflow is reported once (AZ_ERR_ARENA), the games that did not fit are dropped WITHOUT leaving a hole: what comes
    back is exactly the committed games, each a whole, replayable game."""
    Gn = 8
    eng = E.SelfPlayEngine("TTT", n_games=Gn, n_playout=12, c_puct=1.0, evaluator="hash", eval_seed=4, move="sample",
                           rng_seed=9)
    cap = Gn * 9 * 2
    seen_err = 0
    for step in range(60):
        eng.selfplay_step(temp=1.0, first_n_steps=3, refill=True)
        try:
            eng.sync()
        except E.AzError as ex:
            assert "trajectory buffer overflow" in str(ex)
            seen_err += 1
    assert seen_err >= 1
    ng, npl = eng.traj_count()
    assert npl <= cap
    tr = eng.traj_fetch()
    assert len(tr["z"]) == npl and int(tr["done"].sum()) == ng
    lo = 0
    for hi in np.flatnonzero(tr["done"]) + 1:          # every committed game is whole and legal
        env = orc.Env("TTT")
        gid = tr["game_id"][lo]
        for i in range(lo, hi):
            assert tr["game_id"][i] == gid and np.array_equal(tr["states"][i], env.packed())
            env.step(int(tr["action"][i]))
        assert env.done()
        lo = hi
    assert lo == npl
    assert npl > cap - 9        # dropped games left no hole: the buffer is full up to less than one game
    eng.close()


# ------------------------------------------------------------------------------------------------ K1 range checks
def test_env_batch_ignores_out_of_range_actions_and_boards(E):
    """az_env_step_batch / az_env_status_batch with caller-supplied garbage: actions outside [0, A) are no moves (state
    and turn unchanged, placed = 0), an NBTTT state whose current-board nibble is 9..15 has no legal move, is not full,
    and takes no stone."""
    c4 = np.array([orc.Env("Connect4").packed()] * 6)
    out = E.env_step_batch("Connect4", c4, np.array([-1, 7, 100, -2 ** 31, 2 ** 31 - 1, 3], dtype=np.int64).astype(np.int32))
    assert list(out["placed"]) == [0, 0, 0, 0, 0, 1]
    assert np.array_equal(out["states"][:5], c4[:5]) and not np.array_equal(out["states"][5], c4[5])
    ttt = np.array([orc.Env("TTT").packed()] * 3)
    out = E.env_step_batch("TTT", ttt, np.array([-1, 9, 4], dtype=np.int32))
    assert list(out["placed"]) == [0, 0, 1] and np.array_equal(out["states"][:2], ttt[:2])
    nb = np.zeros((8, 4), dtype=np.uint64)
    for i, cur in enumerate([9, 10, 11, 12, 13, 14, 15, 8]):
        nb[i, 3] = cur
    st = E.env_status_batch("NBTTT", nb)
    assert list(st["legal"]) == [0] * 7 + [0x1FF] and list(st["full"]) == [0] * 8 and list(st["winner"]) == [0] * 8
    out = E.env_step_batch("NBTTT", nb, np.full(8, 4, dtype=np.int32))
    assert list(out["placed"]) == [0] * 7 + [1]
    assert np.array_equal(out["states"][:7], nb[:7])


# ------------------------------------------------------------------------------------------------ configs[3] shape
def test_cfg3_shape_16384_slots_rccl_gather_store_many(E, blob):
    """BASELINE configs[3], one GPU's share: 16 384 concurrent Connect4 games x 500 playouts with the real net (38.5 GB
    node pool, 4 cohorts), then finished trajectories packed -> gathered over RCCL (torch.distributed backend "nccl", a
    group of this one rank, collectives forced) -> expanded by store_many into reference-shaped replay tensors.
    Size-independent properties: root N = S and sum(child N) = S - 1 for every slot, identical positions -> identical
    trees, leaf_evals + cache_hits + terminal_hits = playouts, overflow flags clean, gathered records == packed records, every
    stored game a whole game."""
    import torch
    import torch.distributed as dist
    from helpers import RefShapedBuffer
    from alphazero_amd.collector import TrajectoryGatherer, gather_records, store_many
    Gn, S = 16384, 500
    eng = E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, c_puct=1.25, evaluator="cnn", flip="never",
                           reuse_tree=True, move="sample", rng_seed=3)
    assert eng.n_cohorts == 4 and eng.nodes_per_game * 16 * Gn > 38e9
    eng.load_net(blob)
    # late-game roots (two distinct positions, alternating) so that games end within a few plies
    rng = np.random.default_rng(1)
    lates = []
    while len(lates) < 2:
        e = orc.Env("Connect4")
        for _ in range(30):
            if not e.done():
                e.step(int(rng.choice(e.valid_move())))
        if not e.done():
            lates.append(e.packed())
    roots = np.array([lates[g & 1] for g in range(Gn)])
    eng.set_roots(roots)
    eng.search()
    rc = eng.root_children()
    assert (rc["root_visits"] == S).all() and (rc["visits"].sum(1) == S - 1).all()
    for par in (0, 1):      # 8192 identical positions each -> identical trees wherever the leaves sat in the batches
        for k in ("actions", "visits"):
            assert (rc[k][par::2] == rc[k][par]).all(), k
        assert (rc["q"][par::2].view(np.uint32) == rc["q"][par].view(np.uint32)).all()
    st = eng.stats()
    assert st["playouts"] == Gn * S and st["leaf_evals"] + st["cache_hits"] + st["terminal_hits"] == st["playouts"]
    # rolling self-play from these roots until every slot's first game is over (a handful of plies), refill on
    eng.set_roots(roots)
    for _ in range(14):
        eng.selfplay_rounds(S, temp=1.0, first_n_steps=10, refill=True)
        eng.sync()                                      # raises on arena / trajectory overflow
        if eng.traj_count()[0] >= Gn:
            break
    ng, npl = eng.traj_count()
    assert ng >= Gn
    started_here = not dist.is_initialized()
    if started_here:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        local = torch.empty((npl, eng.record_bytes), dtype=torch.uint8, device="cuda")
        assert eng.traj_pack_into(local.data_ptr(), npl, clear=False) == npl
        recs = gather_records(local, force=True)         # counts all_gather + padded all_gather on RCCL
        assert dist.get_backend() == "nccl" and torch.equal(recs, local)
        assert torch.equal(TrajectoryGatherer(eng, torch.device("cuda", 0)).gather(), local)   # the bench's call
        dist.barrier()
    finally:
        if started_here:
            dist.destroy_process_group()
    buf = RefShapedBuffer(3, npl + 7, 7, 6, 7, "cuda")
    assert store_many(buf, recs, "Connect4") == 0 and buf._ptr == npl
    done = buf.done[:npl, 0]
    assert int(done.sum()) == ng and bool(done[-1])
    assert not torch.isnan(buf.state[:npl]).any() and torch.isnan(buf.state[npl:]).all()
    placed = (buf.next_state[:npl, 0] + buf.next_state[:npl, 1]).sum((1, 2)) - (buf.state[:npl, 0] + buf.state[:npl, 1]).sum((1, 2))
    assert (placed == 1).all() and torch.equal(buf.next_state[:npl, 2], -buf.state[:npl, 2])
    z = buf.value[:npl, 0]
    assert torch.equal(z.to(torch.int32), buf.winner[:npl, 0]) and bool(((z == 0) | (z == 1) | (z == -1)).all())
    eng.close()


# ------------------------------------------------------------------------------------------------ configs[4] shape
def test_cfg4_shape_nbttt_deep_trees(E):
    """BASELINE configs[4] on one GPU: nine-board tic-tac-toe, 800 playouts per move, tree reuse, 4 096 games with arenas
    sized for whole games (1 + 9 x 800 x 81 nodes each = 38 GB of node pool).  Properties: overflow flags clean, root N and
    sum(child N) after every ply with reuse (n_a - 1 + S), identical openings -> bit-identical trees, every playout
    either evaluated or terminal."""
    Gn, S = 4096, 800
    eng = E.SelfPlayEngine("NBTTT", n_games=Gn, n_playout=S, c_puct=1.5, evaluator="hash", eval_seed=77, reuse_tree=True,
                           move="first_max")
    assert eng.nodes_per_game == 1 + 9 * S * 81 and eng.nodes_per_game * 16 * Gn > 38e9
    op = []
    for i in range(4):
        op.append(orc.Env.nbttt_open(i, 8 - i, i + 1).packed())
    eng.set_roots(np.array([op[g % 4] for g in range(Gn)]))
    carried = np.zeros(Gn, dtype=np.int64)
    for ply in range(6):
        eng.search()
        eng.sync()
        rc = eng.root_children()
        want_root = np.where(carried > 0, carried + S, S)
        assert np.array_equal(rc["root_visits"], want_root), ply
        assert np.array_equal(rc["visits"].sum(1), want_root - 1), ply
        for c in range(4):
            for k in ("actions", "visits"):
                assert (rc[k][c::4] == rc[k][c]).all(), (ply, k)
            assert (rc["q"][c::4].view(np.uint32) == rc["q"][c].view(np.uint32)).all()
        a, _ = eng.choose_moves(temp=1e-3)
        carried = rc["visits"][np.arange(Gn), (rc["actions"] == a[:, None]).argmax(1)].astype(np.int64)
        eng.advance(a)
    st = eng.stats()
    assert st["playouts"] == 6 * Gn * S and st["leaf_evals"] + st["terminal_hits"] == st["playouts"]
    assert st["nodes_used"] <= Gn * eng.nodes_per_game
    eng.close()


# ------------------------------------------------------------------------------------------------ bench.py launcher
@pytest.mark.parametrize("config", [2, 3, 4])
def test_bench_starts_its_own_ranks_on_one_gpu(tmp_path, config):
    """`python bench.py --gpus N` outside torchrun starts N ranks itself (fresh child processes, RCCL rendezvous on
    127.0.0.1).  One GPU here, so N = 1 with --launch: the child is rank 0 of a 1-rank RCCL group and walks the barrier /
    all-reduce / gather code of the N-GPU job; its line must say n_gpus = 1 = the ranks RCCL reports.  All three first-class
    workloads (BASELINE configs[2], [3], [4]) at sizes the suite can afford."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch", "--steps", "6", "--warmup", "1",
           "--config", str(config), "--games", "256", "--playouts", "40", "--cohort", "64", "--no-cpu-baseline",
           "--no-second-leg", "--preroll", "4"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["value"] >= 0 and d["config"]["plies_stored_in_replay"] >= 0
    assert d["config"]["launcher"] == "bench.py (self-launched ranks)" and d["config"]["rccl_ranks"] == 1
    assert d["config"]["baseline_config"] == config and ("configs[%d]" % config) in d["config"]["workload"]
    assert d["roofline"]["bound"] == ("hbm" if config == 4 else "mfma")
