"""The drop-in mirrors (alphazero_amd/dropin: environments.load / Env / MCTS_AZ / AlphaZeroPlayer / Game) driven the way
the reference's own drivers drive them (play.py:27-58, Pipeline.py:53-63, game.py:35-58), against the golden vectors
that tests/golden/make_fixtures.py recorded from the reference with the SAME harness rules:
np.random.rand -> recorded flip bits, np.random.choice -> first maximum, players in eval() mode.
Visit counts / actions / winners / trajectories: exact.  pi: 1e-6.  Real-net search: move agreement stated."""
import json
import os
import pickle
import sys

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
DROPIN = os.path.join(ROOT, "alphazero_amd", "dropin")


@pytest.fixture(scope="module")
def dropin():
    """what a driver sees after `sys.path.insert(0, <repo>/alphazero_amd/dropin)`."""
    sys.path.insert(0, DROPIN)
    import environments
    import game
    import MCTS
    import player
    yield dict(load=environments.load, game=game, MCTS=MCTS, player=player)
    sys.path.remove(DROPIN)


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(G, "fixtures.json")) as f:
        return json.load(f)


class RngPatch:
    """owns np.random.rand / np.random.choice like the fixture harness (make_fixtures.py RngPatch)."""

    def __init__(self, flip_bits=None):
        self.flip_bits, self.pos = flip_bits, 0

    def rand(self, *a):
        assert not a
        b = 0 if self.flip_bits is None or self.pos >= len(self.flip_bits) else self.flip_bits[self.pos]
        self.pos += 1
        return 0.25 if b else 0.75

    @staticmethod
    def choice(a, p=None, **kw):
        a = list(a)
        return a[0] if p is None else a[int(np.argmax(np.asarray(p)))]

    def __enter__(self):
        self._rand, self._choice = np.random.rand, np.random.choice
        np.random.rand, np.random.choice = self.rand, self.choice
        return self

    def __exit__(self, *exc):
        np.random.rand, np.random.choice = self._rand, self._choice


def _ttt_planes_to_packed(s):
    s = np.asarray(s).reshape(3, 3, 3)
    w = 0
    for i in range(9):
        if s[0, i // 3, i % 3] == 1:
            w |= 1 << i
        if s[1, i // 3, i % 3] == 1:
            w |= 1 << (16 + i)
    if s[2, 0, 0] < 0:
        w |= 1 << 63
    return np.array([w], dtype=np.uint64)


class FakeNet:
    """host-side policy_value_fn with the reference's evaluator interface (.predict/.device/.n_actions, MCTS.py:145-146):
    uniform priors, or the hashed evaluator of DESIGN.md section 2 (here computed by the oracle's C implementation)."""
    device = "cpu"

    def __init__(self, game, kind, seed=0):
        self.game, self.kind, self.seed = game, kind, seed
        self.n_actions = 7 if game == "Connect4" else 9

    def predict(self, state):
        A = self.n_actions
        if self.kind == "uniform":
            return np.full((1, A), np.float32(1.0) / np.float32(A), dtype=np.float32), np.zeros((1, 1), np.float32)
        s = state.numpy()
        if self.game == "Connect4":
            from environments.network import planes_to_packed
            words = planes_to_packed(s)[0]
        else:
            words = _ttt_planes_to_packed(s)
        p, v = orc.eval_hash(orc.Env.from_packed(self.game, words), self.seed)
        return p.reshape(1, A).copy(), np.array([[v]], dtype=np.float32)

    def eval(self):
        pass

    def train(self):
        pass


# ------------------------------------------------------------------------------------------------ Env API
@pytest.mark.parametrize("game", ["Connect4", "TTT", "NBTTT"])
def test_env_api_vs_golden_traces(dropin, game):
    tr = np.load(os.path.join(G, "env_traces.npz"))
    Env = dropin["load"](game).Env
    starts, action = tr[game + "_game_start"], tr[game + "_action"]
    for g in range(12):
        env = None if game == "NBTTT" else Env()
        if game == "NBTTT":  # third opening number picks the second ply among the allowed cells (harness rule)
            b, a1, pick = [int(x) for x in tr["NBTTT_opening"][g]]
            with RngPatch():
                save = np.random.choice
                np.random.choice = lambda v: v[pick % len(v)]
                try:
                    env = Env(b, a1)
                finally:
                    np.random.choice = save
        for i in range(starts[g], starts[g + 1]):
            if action[i] >= 0:
                env.step(int(action[i]))
            cells = env.boards if game == "NBTTT" else env.board
            assert np.array_equal(np.asarray(cells, dtype=np.int8).flatten(), tr[game + "_cells"][i])
            assert env.turn == tr[game + "_turn"][i]
            assert [int(b) for b in env.valid_mask()] == list(tr[game + "_mask"][i])
            assert env.valid_move() == [a for a, b in enumerate(tr[game + "_mask"][i]) if b]
            assert env.winPlayer() == tr[game + "_winner"][i]
            assert env.done() == bool(tr[game + "_done"][i])
            st = env.current_state()
            assert st.dtype == np.float32 and np.array_equal(st.flatten(), tr[game + "_state"][i])


def test_connect4_env_extras(dropin):
    """Env(board) turn inference, full-column no-op, place() keeps the turn, flip / flip_action / random_flip, copy
    independence, pickle (env_cython.pyx:27-57,129-141,168-199)."""
    tr = np.load(os.path.join(G, "env_traces.npz"))
    Env = dropin["load"]("Connect4").Env
    for k in range(40):
        board = tr["c4_rand_cells"][k].reshape(6, 7).astype(np.float32)
        env = Env(board)
        assert env.turn == tr["c4_rand_turn"][k]
        assert np.array_equal(env.board, board)
        assert env.check_winner() == tr["c4_rand_winner"][k]
        assert env.check_full() == bool(tr["c4_rand_full"][k])
        assert [env.flip_action(a) for a in range(7)] == list(tr["c4_rand_flip_action"][k])
        assert np.array_equal(env.flip().board, board[:, ::-1])
        assert [int(b) for b in env.valid_mask()] == list(tr["c4_rand_mask"][k])
        e2 = pickle.loads(pickle.dumps(env))
        assert e2 == env and e2.turn == env.turn
    env = Env()
    for _ in range(6):
        env.step(3)
    before, turn = env.board, env.turn
    env.step(3)  # full column: silent no-op that keeps the turn
    assert np.array_equal(env.board, before) and env.turn == turn
    assert np.array_equal(before.flatten().astype(np.int8), tr["c4_fullcol_after_cells"])
    assert env.place(3) is False and env.place(2) is True and env.turn == turn
    c = env.copy()
    c.step(0)
    assert not np.array_equal(c.board, env.board)
    with RngPatch([1, 0]):
        f, flipped = env.random_flip()
        assert flipped and np.array_equal(f.board, env.board[:, ::-1])
        f, flipped = env.random_flip()
        assert not flipped and f == env
    env.reset()
    assert env.turn == 1 and not env.board.any()


# ------------------------------------------------------------------------------------------------ search + player
CASES = ["c4_hash_s80_flips_reuse_0", "c4_hash_s120_flips_fresh", "c4_uniform_s500_reuse", "ttt_hash_s100_flips_reuse",
         "ttt_hash_s64_flips_fresh_0", "ttt_uniform_s100_reuse", "nbttt_uniform_s800_reuse"]


@pytest.mark.parametrize("name", CASES)
def test_player_get_action_vs_golden(dropin, meta, name):
    """AlphaZeroPlayer.get_action ply by ply with a host evaluator, exactly as make_fixtures.gen_search_case drove the
    reference: (actions, visits) in insertion order, pi and the chosen action must match the reference's."""
    m = meta["search_cases"][name]
    d = np.load(os.path.join(G, "search.npz"))
    game = m["game"]
    net = FakeNet(game, m["kind"], m["seed"])
    player = dropin["player"].AlphaZeroPlayer(net, c_puct=m["c_puct"], n_playout=m["S"], alpha=None,
                                              is_selfplay=1 if m["reuse"] else 0)
    player.eval()
    Env = dropin["load"](game).Env
    flips = d[name + "__flips"] if m["has_flips"] else None
    snap = {}
    orig = player.mcts.get_action_visits

    def spy(e):
        snap["av"] = orig(e)
        snap["root"] = player.mcts.root
        return snap["av"]

    player.mcts.get_action_visits = spy
    plies = len(d[name + "__action"])
    if game == "NBTTT":
        plies = min(plies, 4)   # S=800 through the host evaluator: a few plies are enough here
    with RngPatch(flips):
        if game == "NBTTT":
            b, a1, pick = m["opening"]
            save = np.random.choice
            np.random.choice = lambda v: v[pick % len(v)]
            try:
                env = Env(b, a1)
            finally:
                np.random.choice = save
        else:
            env = Env()
        for ply in range(plies):
            first_n, temp = m["first_n"], m["temp"]
            temperature = (1e-3 if ply >= first_n else temp) if first_n is not None else temp
            action, pi = player.get_action(env, temperature)
            acts, vis = snap["av"]
            k = int(d[name + "__n_children"][ply])
            assert list(acts) == list(d[name + "__actions"][ply][:k]), (name, ply)
            assert list(vis) == list(d[name + "__visits"][ply][:k]), (name, ply)
            root = snap["root"]
            assert root.n_visits == d[name + "__root_N"][ply]
            assert [c.Q.view(np.uint32) for c in root.children.values()] == list(d[name + "__child_Q_bits"][ply][:k])
            assert pi.dtype == np.float32 and pi.shape == (net.n_actions,)
            np.testing.assert_allclose(pi, d[name + "__pi"][ply][:net.n_actions], rtol=1e-6, atol=1e-30)
            assert int(action) == d[name + "__action"][ply], (name, ply)
            env.step(action)
    if plies == len(d[name + "__action"]) and int(d[name + "__winner"]) != 99:
        assert env.done() and env.winPlayer() == int(d[name + "__winner"])


def test_game_start_self_play_vs_golden_trajectory(dropin):
    """Game.start_self_play (game.py:35-58) with the hashed evaluator: the 6-tuples the reference stored."""
    t = np.load(os.path.join(G, "trajectory.npz"))
    net = FakeNet("Connect4", "hash", 2024)
    player = dropin["player"].AlphaZeroPlayer(net, c_puct=1.25, n_playout=60, alpha=None, is_selfplay=1)
    player.eval()
    game = dropin["game"].Game(dropin["load"]("Connect4").Env())
    with RngPatch([int(x) for x in t["flips"]]):
        winner, data = game.start_self_play(player, temp=1.0, first_n_steps=10)
        data = list(data)
    assert winner == int(t["winner"]) and len(data) == len(t["z"])
    for i, (state, pi, z, z2, nxt, done) in enumerate(data):
        assert state.shape == (1, 3, 6, 7) and np.array_equal(state[0], t["state"][i])
        np.testing.assert_allclose(pi, t["pi"][i], rtol=1e-6, atol=1e-30)
        assert z == t["z"][i] and z2 == t["z2"][i]
        assert np.array_equal(nxt[0], t["next_state"][i]) and bool(done) == bool(t["done"][i])


def test_play_py_flow_real_net_on_device(dropin):
    """play.py:27-58 (`--sp -n 500`): load('Connect4'), CNN.load, AlphaZeroPlayer(eval), fresh tree per ply -- the whole
    search on the device (AZ_EVAL_CNN).  Against the 8 recorded reference plies: visit counts may differ where fp32
    rounding of the network differs (1e-4 contract), so the bar is move agreement, which is stated."""
    r = np.load(os.path.join(G, "realnet_search.npz"))
    module = dropin["load"]("Connect4")
    config = module.config.training_config
    env = module.Env()
    net = module.CNN(0, device="cuda")
    net.load(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"))
    az = dropin["player"].AlphaZeroPlayer(net, c_puct=config["c_puct"], n_playout=int(r["S"]),
                                          alpha=config["dirichlet_alpha"], is_selfplay=0)
    az.eval()
    agree = l1 = 0
    with RngPatch([int(x) for x in r["flips"]]):
        for ply in range(len(r["chosen"])):
            acts, vis = az.mcts.get_action_visits(env)
            a = acts[int(np.argmax(vis))]
            k = len(acts)
            assert list(acts) == list(r["actions"][ply][:k])
            l1 += int(np.abs(np.array(vis) - r["visits"][ply][:k]).sum())
            agree += int(a == r["chosen"][ply])
            az.reset_player()
            env.step(int(r["chosen"][ply]))
    print("real-net move agreement %d/%d, visit-count L1 distance %d" % (agree, len(r["chosen"]), l1))
    assert agree == len(r["chosen"])
    # CNN.predict on the reference's own input format
    probs, value = net.predict(module.Env().current_state())
    assert probs.shape == (1, 7) and value.shape == (1, 1)
    np.testing.assert_allclose(probs[0], [0.0240, 0.0494, 0.0632, 0.7262, 0.0630, 0.0478, 0.0265], atol=2e-4)
    np.testing.assert_allclose(value[0, 0], 0.3786, atol=2e-4)


def test_start_play_and_batched_self_play(dropin):
    module = dropin["load"]("Connect4")
    net = module.CNN(0, device="cuda")
    net.load(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"))
    az = dropin["player"].AlphaZeroPlayer(net, c_puct=1.25, n_playout=48, alpha=0.7, is_selfplay=0)
    az.eval()
    np.random.seed(3)
    winner = dropin["game"].Game(module.Env()).start_play(az, az, show=0)
    assert winner in (-1, 0, 1)
    # Pipeline.data_collector's configuration (train mode, tree reuse), 32 games at once on the device
    sp = dropin["player"].AlphaZeroPlayer(net, c_puct=1.25, n_playout=32, alpha=0.7, is_selfplay=1)
    sp.train()
    games = dropin["game"].Game(module.Env()).start_self_play_batch(sp, 32, temp=1.0, first_n_steps=10)
    assert len(games) == 32
    for winner, data in games:
        assert 7 <= len(data) <= 42 and data[-1][5] and not any(d[5] for d in data[:-1])
        last = module.Env()
        for state, pi, z, z2, nxt, done in data:
            assert state.shape == (1, 3, 6, 7) and abs(float(pi.sum()) - 1) < 1e-5 and z == z2
            mover = 1 if state[0, 2, 0, 0] > 0 else -1
            assert z == (0 if winner == 0 else (1 if mover == winner else -1))
            assert np.array_equal(state, last.current_state())
            diff = (nxt[0, 0] + nxt[0, 1]) - (state[0, 0] + state[0, 1])
            assert diff.sum() == 1
            last.step(int(np.argmax(diff.sum(0))))
            assert np.array_equal(nxt, last.current_state())
        assert last.done() and last.winPlayer() == winner


def test_collect_self_play_into_reference_shaped_buffer(dropin):
    """Pipeline.data_collector's job in one call: n games on the device, plies written into a ReplayBuffer-shaped object
    on the GPU by the unpack kernel; lengths, terminal flags and outcome labels are consistent."""
    import torch
    from helpers import RefShapedBuffer as _RefShapedBuffer
    module = dropin["load"]("Connect4")
    net = module.CNN(0, device="cuda")
    net.load(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"))
    sp = dropin["player"].AlphaZeroPlayer(net, c_puct=1.25, n_playout=24, alpha=0.7, is_selfplay=1)
    sp.train()
    buf = _RefShapedBuffer(3, 4096, 7, 6, 7, torch.device("cuda", 0))
    lens = dropin["game"].Game(module.Env()).collect_self_play(sp, buf, 40, temp=1.0, first_n_steps=10)
    assert len(lens) == 40 and all(7 <= n <= 42 for n in lens) and buf._ptr == sum(lens)
    n = buf._ptr
    done = buf.done[:n, 0].cpu().numpy()
    assert done.sum() == 40 and list(np.flatnonzero(done) + 1) == list(np.cumsum(lens))
    assert not torch.isnan(buf.state[:n]).any() and not torch.isnan(buf.next_state[:n]).any()
    assert torch.isnan(buf.state[n:]).all()
    np.testing.assert_allclose(buf.prob[:n].sum(1).cpu().numpy(), 1.0, rtol=1e-5)
    z = buf.value[:n, 0].cpu().numpy()
    assert set(np.unique(z)) <= {-1.0, 0.0, 1.0} and np.array_equal(z.astype(np.int32), buf.winner[:n, 0].cpu().numpy())
    lo = 0
    for ln in lens:   # inside a game the outcome label alternates with the mover (game.py:52-55)
        zz = z[lo:lo + ln]
        assert (zz[::-1][0::2] == zz[-1]).all() and (zz[::-1][1::2] == -zz[-1]).all()
        lo += ln


def test_dropin_replaybuffer_collects_and_serves_batches_on_device(dropin):
    """SURVEY 8f-1 end to end with the drop-in's own ReplayBuffer on the GPU: collect_self_play -> store_many (unpack
    kernel, ring wrap at current_capacity) -> len without a NaN scan -> dataloader / sample_balanced / sample hand out
    device-resident rows that are whole plies; the same rows land where per-ply store() puts them."""
    import torch
    from ReplayBuffer import ReplayBuffer
    from alphazero_amd.collector import TrajectoryGatherer
    from alphazero_amd.engine import SelfPlayEngine
    module = dropin["load"]("Connect4")
    net = module.CNN(0, device="cuda")
    net.load(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"))
    sp = dropin["player"].AlphaZeroPlayer(net, c_puct=1.25, n_playout=24, alpha=0.7, is_selfplay=1)
    sp.train()
    buf = ReplayBuffer(3, 4096, 7, 6, 7, replay_ratio=0.25, device="cuda")
    lens = dropin["game"].Game(module.Env()).collect_self_play(sp, buf, 64, temp=1.0, first_n_steps=10)
    n = sum(lens)
    assert len(buf) == n == buf._ptr and buf.recount() == n and not buf.is_full()
    torch.manual_seed(0)
    dl = buf.dataloader(128)
    rows = dl.dataset.tensors
    budget = n if n <= 1000 else min(int(n * 0.25), 10000)
    quota = max(1, int(budget * 0.2))                                # game-ending plies asked for (ReplayBuffer.py:108)
    n_end = int(rows[5].sum().item())
    end_w = buf.winner[:n][buf.done[:n]].reshape(-1)
    classes = torch.unique(end_w)
    want_end = sum(min(int((end_w == v).sum()), quota // len(classes) + (1 if i < quota % len(classes) else 0))
                   for i, v in enumerate(classes))
    assert all(t.is_cuda for t in rows) and n_end == want_end and rows[0].shape[0] == n_end + budget - quota
    assert len(torch.unique(rows[0].reshape(rows[0].shape[0], -1), dim=0)) >= 0.5 * rows[0].shape[0]
    state, prob, value, winner, nxt, done = next(iter(dl))
    assert state.shape[1:] == (3, 6, 7) and prob.shape[1:] == (7,) and done.dtype == torch.bool
    assert not torch.isnan(state).any() and torch.equal(value.to(torch.int32), winner)
    placed = (nxt[:, 0] + nxt[:, 1]).sum((1, 2)) - (state[:, 0] + state[:, 1]).sum((1, 2))
    assert (placed == 1).all()                                     # next_state = state + the move that was played
    assert torch.equal(nxt[:, 2], -state[:, 2])                    # and the turn plane changes sign
    s6 = buf.sample(33)
    assert s6[0].shape == (33, 3, 6, 7) and s6[0].is_cuda
    bl = buf.sample_balanced(64)
    w, allw = bl.dataset.tensors[3].reshape(-1), buf.winner[:n].reshape(-1)
    total = min(n, 1024)                                             # min(len, replay_ratio * allocated rows, 10000)
    classes = torch.unique(allw)
    for i, v in enumerate(classes):                                  # outcomes evenly represented, as far as they exist
        q = total // len(classes) + (1 if i < total % len(classes) else 0)
        assert int((w == v).sum()) == min(q, int((allw == v).sum()))
    # store_many == a loop of store(): same rows, same pointer, ring wrap included
    eng = SelfPlayEngine("Connect4", n_games=16, n_playout=20, evaluator="hash", eval_seed=9, move="first_max")
    eng.selfplay(16, temp=1.0, first_n_steps=4)
    tr = eng.traj_fetch(clear=False)
    recs = TrajectoryGatherer(eng, torch.device("cuda", 0)).gather()
    m = len(tr["z"])
    a = ReplayBuffer(3, m + 50, 7, 6, 7, device="cuda")
    b = ReplayBuffer(3, m + 50, 7, 6, 7, device="cuda")
    a.current_capacity = b.current_capacity = m - 9
    a._ptr = b._ptr = 4
    assert a.store_many(recs, "Connect4") == 4
    for i in range(m):
        env = orc.Env.from_packed("Connect4", tr["states"][i])
        s = env.current_state()
        env.step(int(tr["action"][i]))
        b.store(s, tr["pi"][i], float(tr["z"][i]), float(tr["z"][i]), env.current_state(), bool(tr["done"][i]))
    assert (len(a), a._ptr) == (len(b), b._ptr) == (m - 9, (4 + m) % (m - 9))
    for name in ("state", "prob", "value", "winner", "next_state", "done"):
        x, y = getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy()
        assert np.array_equal(x, y, equal_nan=x.dtype.kind == "f"), name
    eng.close()
