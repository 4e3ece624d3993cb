#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Container-only (needs /root/reference and oracle/_ref/env_cython.so from oracle/build_ref.sh).
The reference ships no tests, fixtures or golden vectors, so every parity
claim of this repo is anchored on the outputs recorded here.  Only data is written: inputs and the
reference's outputs -- never reference source text.

    python tests/golden/make_fixtures.py            # regenerate everything (about 3 minutes)

Determinism: the reference never seeds np.random; the harness owns all randomness.
  * np.random.rand   (env.random_flip, one draw per playout)  -> replaced by a recorded bit stream
  * np.random.choice (player.get_action, one draw per ply)    -> replaced by "first maximum of p"
  * np.random.dirichlet: never reached (players are put in eval() mode)
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload  # noqa: E402

ref = _refload.load_reference()
M64 = (1 << 64) - 1


# ----------------------------------------------------------------------------- hashed evaluator
def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def pack_planes(game, s):
    """state planes (as the evaluator sees them) -> packed words of include/azamd.h"""
    s = np.asarray(s)
    if game == "Connect4":
        x, o, t = s[0, 0], s[0, 1], s[0, 2, 0, 0]
        w0 = w1 = 0
        for r in range(6):
            for c in range(7):
                bit = c * 7 + (5 - r)
                if x[r, c] == 1:
                    w0 |= 1 << bit
                if o[r, c] == 1:
                    w1 |= 1 << bit
        if t != 1:
            w1 |= 1 << 63
        return [w0, w1]
    if game == "TTT":
        x, o, t = s[0, 0], s[0, 1], s[0, 2, 0, 0]
        w = 0
        for i in range(9):
            if x.flat[i] == 1:
                w |= 1 << i
            if o.flat[i] == 1:
                w |= 1 << (16 + i)
        if t != 1:
            w |= 1 << 63
        return [w]
    w = [0, 0, 0, 0]
    for b in range(9):
        for k in range(9):
            sh = 18 * (b % 3)
            if s[0, b].flat[k] == 1:
                w[b // 3] |= 1 << (sh + k)
            if s[0, 9 + b].flat[k] == 1:
                w[b // 3] |= 1 << (sh + 9 + k)
    curr = [b for b in range(9) if s[0, 18 + b, 0, 0] == 1][0]
    w[3] = curr
    if s[0, 27, 0, 0] != 1:
        w[3] |= 1 << 63
    return w


def hash_eval(game, words, seed, A):
    h = seed
    for w in words:
        h = mix64(h ^ w)
    raw = [np.float32((mix64((h + a + 1) & M64) >> 40) + 1) * np.float32(1.0 / 16777216.0) for a in range(A)]
    s = np.float32(0)
    for r in raw:
        s = np.float32(s + r)
    probs = np.array([np.float32(r / s) for r in raw], dtype=np.float32)
    value = np.float32(np.float32(mix64((h + 64) & M64) >> 40) * np.float32(1.0 / 8388608.0) - np.float32(1.0))
    return probs, value


class FakeNet:
    """Stand-in for CNN (Network.py:30-85): .predict returns np.float32 (1,A) and (1,1) like the real one."""
    device = "cpu"

    def __init__(self, game, kind, seed=0):
        self.game, self.kind, self.seed = game, kind, seed
        self.n_actions = 7 if game == "Connect4" else 9
        self.calls = 0

    def predict(self, state):
        self.calls += 1
        A = self.n_actions
        if self.kind == "uniform":
            p = np.full((1, A), np.float32(1.0) / np.float32(A), dtype=np.float32)
            return p, np.zeros((1, 1), dtype=np.float32)
        probs, value = hash_eval(self.game, pack_planes(self.game, state.numpy()), self.seed, A)
        return probs.reshape(1, A).copy(), np.array([[value]], dtype=np.float32)

    def eval(self):
        pass

    def train(self):
        pass


# ----------------------------------------------------------------------------- RNG control
class RngPatch:
    """Owns np.random.rand / np.random.choice while the reference runs."""

    def __init__(self, flip_bits=None):
        self.flip_bits = flip_bits  # None -> never flip
        self.pos = 0
        self.choices = 0

    def rand(self, *a):
        assert not a
        if self.flip_bits is None:
            self.pos += 1
            return 1.0
        b = self.flip_bits[self.pos] if self.pos < len(self.flip_bits) else 0
        self.pos += 1
        return 0.25 if b else 0.75  # random_flip(p=0.5): flipped iff rand() < 0.5

    def choice(self, a, p=None, **kw):
        self.choices += 1
        a = list(a)
        if p is None:
            return a[0]
        return a[int(np.argmax(np.asarray(p)))]

    def __enter__(self):
        self._rand, self._choice = np.random.rand, np.random.choice
        np.random.rand, np.random.choice = self.rand, self.choice
        return self

    def __exit__(self, *exc):
        np.random.rand, np.random.choice = self._rand, self._choice


# ----------------------------------------------------------------------------- NBTTT harness
def load_nbttt():
    """Rule functions of environments/NBTTT (utils.py:136-268, env.py:29-60) + the C agent's terminal rule.

    The package __init__ shells out to gcc inside the (read-only) reference tree and its link line lacks
    -fPIC, so the package is registered without running __init__, and NBTTTAgent.so is compiled from the
    reference's own C sources into oracle/_ref/ (cwd-relative path expected by NBTTT/utils.py:16)."""
    import importlib
    import subprocess
    import types
    refdir = os.path.join(_refload.REF, "environments", "NBTTT")
    outroot = os.path.join(_refload.REPO, "oracle", "_ref", "cwd")
    outdir = os.path.join(outroot, "environments", "NBTTT")
    os.makedirs(outdir, exist_ok=True)
    so = os.path.join(outdir, "NBTTTAgent.so")
    if not os.path.exists(so):
        srcs = [os.path.join(refdir, "dependency", f) for f in ("Negamax.c", "heuristic.c", "utils.c")]
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-w", "-o", so] + srcs + ["-lm"])
    pkg = types.ModuleType("environments.NBTTT")
    pkg.__path__ = [refdir]
    sys.modules["environments.NBTTT"] = pkg
    cwd = os.getcwd()
    os.chdir(outroot)
    try:
        utils = importlib.import_module("environments.NBTTT.utils")
        envmod = importlib.import_module("environments.NBTTT.env")
    finally:
        os.chdir(cwd)
    import ctypes
    clib = ctypes.CDLL(so)

    class Env(envmod.Env):
        """Harness-side completion of the unfinished reference Env."""

        def __init__(self, board=0, a1=0, pick=0):
            save = np.random.choice
            np.random.choice = lambda v: v[pick % len(v)]
            try:
                super().__init__(board, a1)
            finally:
                np.random.choice = save

        def copy(self):
            e = Env.__new__(Env)
            e.boards = self.boards.copy()
            e.turn, e.curr, e.action, e.count = self.turn, self.curr, self.action, self.count
            return e

        def valid_move(self):
            return [int(a) for a in super().valid_move()]

        def valid_mask(self):
            v = self.valid_move()
            return [a in v for a in range(9)]

        def check_full(self):  # build-defined: current sub-board full (Negamax.c:34 / utils.c:88-97)
            arr = (ctypes.c_int * 100)(*[int(x) for x in self.boards.flatten()])
            mat = ctypes.cast(arr, ctypes.POINTER(ctypes.c_int * 10))
            return bool(clib.is_full(mat, self.curr + 1))

        def check_full_reference(self):  # NBTTT/env.py:49-50 (all 81 cells)
            return envmod.Env.check_full(self)

        def current_state(self):  # build-defined plane indexing (utils.py:258-268 is off by one)
            return utils.board_to_state(self.curr, self.boards, self.turn)

        def random_flip(self, p=0.5):
            return self.copy(), False

        def flip_action(self, a):
            return a

    return Env, utils, clib


# ----------------------------------------------------------------------------- A. env traces
def cells_of(game, env):
    if game == "NBTTT":
        return np.asarray(env.boards, dtype=np.int8).flatten()
    return np.asarray(env.board, dtype=np.int8).flatten()


def gen_env_traces(rng, NB):
    out = {}
    for game, n_games in (("Connect4", 120), ("TTT", 150), ("NBTTT", 120)):
        A = 7 if game == "Connect4" else 9
        rec = dict(game_start=[], action=[], cells=[], turn=[], mask=[], winner=[], done=[], state=[], curr=[],
                   full81=[], opening=[])
        for g in range(n_games):
            if game == "Connect4":
                env = ref.C4.Env()
            elif game == "TTT":
                env = ref.TTT.Env()
            else:
                op = (int(rng.integers(9)), int(rng.integers(9)), int(rng.integers(9)))
                env = NB(*op)
                rec["opening"].append(op)
            rec["game_start"].append(len(rec["action"]))
            # row 0 of every game = the initial position (action -1)
            first = True
            while True:
                if first:
                    a = -1
                    first = False
                else:
                    a = int(rng.choice(env.valid_move()))
                    env.step(a)
                rec["action"].append(a)
                rec["cells"].append(cells_of(game, env))
                rec["turn"].append(env.turn)
                rec["mask"].append(np.array(env.valid_mask(), dtype=np.uint8))
                rec["winner"].append(int(env.winPlayer()))
                rec["done"].append(bool(env.done()))
                rec["state"].append(np.asarray(env.current_state(), dtype=np.float32).flatten())
                if game == "NBTTT":
                    rec["curr"].append(env.curr)
                    rec["full81"].append(bool(env.check_full_reference()))
                if env.done():
                    break
        rec["game_start"].append(len(rec["action"]))
        for k, v in rec.items():
            if len(v):
                out["%s_%s" % (game, k)] = np.array(v)
    # --- Connect4 edge cases (env_cython.pyx) -----------------------------------------------
    # (1) step on a full column: silent no-op, turn unchanged (:129-141)
    env = ref.C4.Env()
    for _ in range(6):
        env.step(3)
    before = (cells_of("Connect4", env).copy(), env.turn)
    env.step(3)
    out["c4_fullcol_before_cells"], out["c4_fullcol_before_turn"] = before[0], before[1]
    out["c4_fullcol_after_cells"], out["c4_fullcol_after_turn"] = cells_of("Connect4", env), env.turn
    # (2) random boards (not necessarily reachable): Env(board) turn inference (:52-57), scan-order winner
    #     (:79-115), flip_action (:174-190), valid_move, check_full, flip
    boards, turns, winners, fulls, flipacts, masks = [], [], [], [], [], []
    for i in range(600):
        fill = rng.random()
        b = np.zeros((6, 7), dtype=np.float32)
        for c in range(7):
            h = int(rng.integers(0, 7)) if fill < 0.8 else 6
            col = rng.choice([1.0, -1.0], size=h)
            for j in range(h):
                b[5 - j, c] = col[j]
        if i % 5 == 0:  # mirror-symmetric boards
            b[:, 4:] = b[:, :3][:, ::-1]
        env = ref.C4.Env(b)
        boards.append(b.astype(np.int8).flatten())
        turns.append(env.turn)
        winners.append(env.winPlayer())
        fulls.append(env.check_full())
        flipacts.append([env.flip_action(a) for a in range(7)])
        masks.append(np.array(env.valid_mask(), dtype=np.uint8))
    out["c4_rand_cells"], out["c4_rand_turn"] = np.array(boards), np.array(turns)
    out["c4_rand_winner"], out["c4_rand_full"] = np.array(winners), np.array(fulls)
    out["c4_rand_flip_action"], out["c4_rand_mask"] = np.array(flipacts), np.array(masks)
    # (3) TTT random boards: winner scan order (TTT/utils.py:17-30), flip_action (TTT/env.py:67-69)
    boards, winners = [], []
    for i in range(400):
        b = rng.choice([0.0, 1.0, -1.0], size=(3, 3)).astype(np.float32)
        env = ref.TTT.Env()
        env.board = b.copy()
        boards.append(b.astype(np.int8).flatten())
        winners.append(int(env.winPlayer()))
    out["ttt_rand_cells"], out["ttt_rand_winner"] = np.array(boards), np.array(winners)
    out["ttt_flip_action"] = np.array([ref.TTT.Env().flip_action(a) for a in range(9)])
    # (4) NBTTT random boards: winner order (NBTTT/utils.py:178-201,238-245)
    boards, winners = [], []
    for i in range(400):
        e = NB(0, 0, 0)
        bb = np.full((10, 10), 2, dtype=np.int64)
        bb[1:, 1:] = rng.choice([2, 0, 1], size=(9, 9), p=[0.4, 0.3, 0.3])
        e.boards = bb
        boards.append(bb.astype(np.int8).flatten())
        winners.append(int(e.winPlayer()))
    out["nbttt_rand_cells"], out["nbttt_rand_winner"] = np.array(boards), np.array(winners)
    return out


# ----------------------------------------------------------------------------- B. PUCT snapshots
def gen_puct(rng):
    TreeNode = ref.MCTS.TreeNode
    n = 20000
    c_init = rng.choice([1.0, 1.25, 1.5, 4.0, 1.1, 2.7], size=n)
    Np = rng.integers(1, 5000, size=n)
    nn = np.minimum(rng.integers(0, 3000, size=n), Np)
    nn[rng.random(n) < 0.05] = 0
    Q = (rng.random(n) * 2 - 1).astype(np.float32)
    Q[rng.random(n) < 0.1] = rng.choice([-1.0, 0.0, 1.0])
    P = rng.random(n).astype(np.float32)
    score = np.zeros(n, dtype=np.float32)
    for i in range(n):
        parent = TreeNode(None, 1)
        parent.n_visits = int(Np[i])
        ch = TreeNode(parent, np.float32(P[i]))
        ch.n_visits = int(nn[i])
        ch.Q = np.float32(Q[i]) if nn[i] > 0 else 0
        r = ch.PUCT(float(c_init[i]), 500)
        score[i] = np.float32(r)
    return dict(c_init=c_init.astype(np.float64), Np=Np.astype(np.int32), n=nn.astype(np.int32), Q=Q, P=P,
                score_bits=score.view(np.uint32))


# ----------------------------------------------------------------------------- C. search
def new_env(game, NB, opening=None):
    if game == "Connect4":
        return ref.C4.Env()
    if game == "TTT":
        return ref.TTT.Env()
    return NB(*opening)


def gen_search_case(game, NB, kind, seed, S, c_puct, flips, reuse, plies, temp, first_n, opening=None, nodes=False):
    """Drive the reference AlphaZeroPlayer ply by ply; record (actions, visits, pi, action) per ply."""
    net = FakeNet(game, kind, seed)
    A = net.n_actions
    player = ref.player.AlphaZeroPlayer(net, c_puct=c_puct, n_playout=S, alpha=None, is_selfplay=1 if reuse else 0)
    player.eval()
    env = new_env(game, NB, opening)
    rec = dict(actions=[], visits=[], pi=[], action=[], root_N=[], root_Q_bits=[], child_Q_bits=[], child_P_bits=[],
               n_children=[])
    snap = {}
    orig_gav = player.mcts.get_action_visits

    def spy(e):  # observe the tree after the reference's own search, before player.get_action prunes it
        acts, vis = orig_gav(e)
        acts, vis = tuple(acts), tuple(vis)
        root = player.mcts.root
        snap.update(acts=list(acts), vis=list(vis), root_N=root.n_visits, root_Q=np.float32(root.Q),
                    cq=[np.float32(root.children[a].Q) for a in acts],
                    cp=[np.float32(root.children[a].prior) for a in acts])
        return acts, vis

    player.mcts.get_action_visits = spy
    with RngPatch(flips) as rp:
        for ply in range(plies):
            if env.done():
                break
            temperature = (1e-3 if ply >= first_n else temp) if first_n is not None else temp
            action, pi_full = player.get_action(env, temperature)  # player.py:101-116, unmodified
            action = int(action)
            acts, vis = snap["acts"], snap["vis"]
            pad = 9 - len(acts)
            rec["actions"].append(acts + [-1] * pad)
            rec["visits"].append(vis + [0] * pad)
            rec["n_children"].append(len(acts))
            rec["pi"].append(np.concatenate([pi_full, np.zeros(9 - A, dtype=np.float32)]))
            rec["action"].append(action)
            rec["root_N"].append(snap["root_N"])
            rec["root_Q_bits"].append(snap["root_Q"].view(np.uint32))
            rec["child_Q_bits"].append([q.view(np.uint32) for q in snap["cq"]] + [0] * pad)
            rec["child_P_bits"].append([q.view(np.uint32) for q in snap["cp"]] + [0] * pad)
            env.step(action)
        n_rand = rp.pos
    out = {k: np.array(v) for k, v in rec.items()}
    out["winner"] = np.array(env.winPlayer() if env.done() else 99)
    out["n_rand"] = np.array(n_rand)
    return out


def gen_search(rng, NB):
    cases = {}
    meta = {}

    def add(name, **kw):
        flips = kw.get("flips")
        r = gen_search_case(**{k: v for k, v in kw.items()}, NB=NB)
        for k, v in r.items():
            cases["%s__%s" % (name, k)] = v
        if flips is not None:
            cases["%s__flips" % name] = np.array(flips, dtype=np.uint8)
        m = {k: v for k, v in kw.items() if k != "flips"}
        m["has_flips"] = flips is not None
        meta[name] = m
        print("  search case", name, "plies", len(r["action"]), "rand draws", int(r["n_rand"]))

    fl = lambda n: [int(x) for x in rng.integers(0, 2, size=n)]
    # Connect4
    add("c4_uniform_s500_reuse", game="Connect4", kind="uniform", seed=0, S=500, c_puct=1.25, flips=None, reuse=True,
        plies=4, temp=1.0, first_n=0)
    add("c4_hash_s500_flips_reuse", game="Connect4", kind="hash", seed=12345, S=500, c_puct=1.25, flips=fl(500 * 48),
        reuse=True, plies=48, temp=1.0, first_n=10)
    add("c4_hash_s200_noflip_fresh", game="Connect4", kind="hash", seed=777, S=200, c_puct=1.25, flips=None,
        reuse=False, plies=48, temp=0, first_n=None)
    add("c4_hash_s120_flips_fresh", game="Connect4", kind="hash", seed=99, S=120, c_puct=1.5, flips=fl(120 * 48),
        reuse=False, plies=48, temp=0, first_n=None)
    for i in range(6):
        add("c4_hash_s80_flips_reuse_%d" % i, game="Connect4", kind="hash", seed=1000 + i, S=80, c_puct=1.25,
            flips=fl(80 * 48), reuse=True, plies=48, temp=1.0, first_n=10)
    # TTT
    add("ttt_uniform_s100_reuse", game="TTT", kind="uniform", seed=0, S=100, c_puct=1.0, flips=None, reuse=True,
        plies=9, temp=1.0, first_n=1)
    add("ttt_hash_s100_flips_reuse", game="TTT", kind="hash", seed=4242, S=100, c_puct=1.0, flips=fl(100 * 9),
        reuse=True, plies=9, temp=1.0, first_n=1)
    for i in range(6):
        add("ttt_hash_s64_flips_fresh_%d" % i, game="TTT", kind="hash", seed=50 + i, S=64, c_puct=1.0,
            flips=fl(64 * 9), reuse=False, plies=9, temp=0, first_n=None)
    # NBTTT (never flips)
    add("nbttt_uniform_s800_reuse", game="NBTTT", kind="uniform", seed=0, S=800, c_puct=1.5, flips=None, reuse=True,
        plies=12, temp=1.0, first_n=12, opening=(4, 4, 0))
    add("nbttt_hash_s800_reuse", game="NBTTT", kind="hash", seed=31337, S=800, c_puct=1.5, flips=None, reuse=True,
        plies=80, temp=1.0, first_n=12, opening=(2, 7, 3))
    for i in range(4):
        add("nbttt_hash_s100_reuse_%d" % i, game="NBTTT", kind="hash", seed=600 + i, S=100, c_puct=1.5, flips=None,
            reuse=True, plies=80, temp=1.0, first_n=12, opening=(i, 8 - i, i + 1))
    return cases, meta


# ----------------------------------------------------------------------------- D. CNN
def export_blob(sd):
    parts = [v.detach().cpu().numpy().astype(np.float32).ravel() for k, v in sd.items() if v.dtype == torch.float32]
    return np.concatenate(parts)


def gen_cnn(rng):
    torch.set_num_threads(1)
    net = ref.C4.CNN(0, device="cpu")
    path = os.path.join(_refload.REF, "params", "AZ_Connect4_CNN_current.pt")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    net.load_state_dict(sd)
    net.eval()
    blob = export_blob(sd)
    cells, turns, states = [], [], []
    while len(cells) < 256:
        env = ref.C4.Env()
        depth = int(rng.integers(0, 36))
        for _ in range(depth):
            if env.done():
                break
            env.step(int(rng.choice(env.valid_move())))
        if env.done():
            continue
        if len(cells) % 2 == 1:
            env = env.flip()
        cells.append(cells_of("Connect4", env))
        turns.append(env.turn)
        states.append(env.current_state()[0])
    st = torch.from_numpy(np.array(states)).float()
    with torch.no_grad():
        logp, logv = net.forward(st)
    probs, value = net.predict(st)
    out = dict(cells=np.array(cells), turn=np.array(turns), logp=logp.numpy(), logv=logv.numpy(), probs=probs,
               value=value)
    # batch-1 calls (what MCTS does) must agree with the batched call to well below the tolerance
    p1, v1 = net.predict(st[:1])
    assert np.allclose(p1, probs[:1], rtol=1e-5, atol=1e-7)
    return out, blob, net


# ----------------------------------------------------------------------------- E. move selection
def gen_moves(rng):
    visits, temps, pis, acts, nch = [], [], [], [], []
    net = FakeNet("Connect4", "uniform")

    class StubMCTS:
        def __init__(self):
            self.ret = None

        def get_action_visits(self, env):
            return self.ret

        def prune_root(self, a):
            pass

    for i in range(600):
        A = 7 if i % 2 == 0 else 9
        k = int(rng.integers(1, A + 1))
        actions = sorted(rng.choice(A, size=k, replace=False).tolist())
        if rng.random() < 0.5:
            actions = actions[::-1]
        v = rng.integers(1, 500, size=k)
        if rng.random() < 0.4 and k > 1:
            v[rng.integers(k)] = v.max()  # ties
        T = [0, 1e-3, 1.0][i % 3]
        net.n_actions = A
        player = ref.player.AlphaZeroPlayer(net, 1.0, 1, None, is_selfplay=1)
        player.mcts = StubMCTS()
        player.mcts.ret = (tuple(actions), tuple(int(x) for x in v))
        with RngPatch(None):
            action, pi = player.get_action(None, T)
        visits.append(list(v) + [0] * (9 - k))
        acts.append(actions + [-1] * (9 - k))
        nch.append(k)
        temps.append(T)
        pis.append(np.concatenate([pi, np.zeros(9 - A, dtype=np.float32)]))
    return dict(visits=np.array(visits), actions=np.array(acts), n=np.array(nch), temp=np.array(temps),
                pi=np.array(pis), n_actions=np.array([7 if i % 2 == 0 else 9 for i in range(600)]),
                chosen=np.array([0] * 0))


def gen_moves_with_choice(rng):
    d = gen_moves(rng)
    # chosen action under the harness rule (first maximum of the sampling distribution)
    chosen = []
    for i in range(len(d["n"])):
        k = int(d["n"][i])
        v = d["visits"][i][:k]
        a = d["actions"][i][:k]
        T = d["temp"][i]
        if T == 0:
            chosen.append(int(a[int(np.argmax(v))]))
        else:
            p = ref.utils.softmax(np.log(v) / max(T, 1e-8))
            chosen.append(int(a[int(np.argmax(p))]))
    d["chosen"] = np.array(chosen)
    return d


# ----------------------------------------------------------------------------- F. trajectories
def gen_trajectory(rng):
    """game.py:35-58 through the reference Game with a hashed evaluator; the 6-tuples as stored."""
    net = FakeNet("Connect4", "hash", 2024)
    player = ref.player.AlphaZeroPlayer(net, c_puct=1.25, n_playout=60, alpha=None, is_selfplay=1)
    player.eval()
    game = ref.game.Game(ref.C4.Env())
    flips = [int(x) for x in rng.integers(0, 2, size=60 * 48)]
    with RngPatch(flips):
        winner, data = game.start_self_play(player, temp=1.0, first_n_steps=10)
        data = list(data)
    return dict(flips=np.array(flips, dtype=np.uint8), winner=np.array(winner),
                state=np.array([d[0][0] for d in data]), pi=np.array([d[1] for d in data]),
                z=np.array([d[2] for d in data], dtype=np.float32), z2=np.array([d[3] for d in data], dtype=np.float32),
                next_state=np.array([d[4][0] for d in data]), done=np.array([d[5] for d in data]))


# ----------------------------------------------------------------------------- G. real-net search
def gen_realnet(net, rng):
    """BASELINE config 1 head: play.py --sp semantics (eval, temp 0, fresh tree per ply), S=500, real CNN."""
    S, plies = 500, 8
    flips = [int(x) for x in rng.integers(0, 2, size=S * plies)]
    player = ref.player.AlphaZeroPlayer(net, c_puct=1.25, n_playout=S, alpha=0.7, is_selfplay=0)
    player.eval()
    env = ref.C4.Env()
    acts_l, vis_l, chosen = [], [], []
    with RngPatch(flips):
        for ply in range(plies):
            acts, vis = player.mcts.get_action_visits(env)
            acts, vis = list(acts), list(vis)
            a = acts[int(np.argmax(vis))]
            acts_l.append(acts + [-1] * (7 - len(acts)))
            vis_l.append(vis + [0] * (7 - len(vis)))
            chosen.append(a)
            player.reset_player()
            env.step(a)
    return dict(flips=np.array(flips, dtype=np.uint8), actions=np.array(acts_l), visits=np.array(vis_l),
                chosen=np.array(chosen), S=np.array(S))


def main():
    rng = np.random.default_rng(20261110)
    NB, nb_utils, nb_clib = load_nbttt()
    header = dict(numpy=np.__version__, torch=torch.__version__, python=sys.version.split()[0],
                  reference="Sunshine-718/AlphaZero @ /root/reference (2025-08-24 snapshot)",
                  note="outputs of the reference itself; see make_fixtures.py for the harness rules")
    print("env traces ...")
    np.savez_compressed(os.path.join(HERE, "env_traces.npz"), **gen_env_traces(rng, NB))
    print("puct ...")
    np.savez_compressed(os.path.join(HERE, "puct.npz"), **gen_puct(rng))
    print("moves ...")
    np.savez_compressed(os.path.join(HERE, "moves.npz"), **gen_moves_with_choice(rng))
    print("trajectory ...")
    np.savez_compressed(os.path.join(HERE, "trajectory.npz"), **gen_trajectory(rng))
    print("cnn ...")
    cnn, blob, net = gen_cnn(rng)
    np.savez_compressed(os.path.join(HERE, "cnn.npz"), **cnn)
    wdir = os.path.join(HERE, "weights")
    os.makedirs(wdir, exist_ok=True)
    blob.astype("<f4").tofile(os.path.join(wdir, "AZ_Connect4_CNN_current.f32"))
    header["weights"] = dict(file="weights/AZ_Connect4_CNN_current.f32", floats=int(blob.size),
                             sha256=hashlib.sha256(blob.astype("<f4").tobytes()).hexdigest(),
                             source="params/AZ_Connect4_CNN_current.pt (MIT, see weights/LICENSE.reference)",
                             layout="all float32 state_dict tensors concatenated in state_dict order")
    print("real-net search ...")
    np.savez_compressed(os.path.join(HERE, "realnet_search.npz"), **gen_realnet(net, rng))
    print("search ...")
    cases, meta = gen_search(rng, NB)
    np.savez_compressed(os.path.join(HERE, "search.npz"), **cases)
    header["search_cases"] = meta
    with open(os.path.join(HERE, "fixtures.json"), "w") as f:
        json.dump(header, f, indent=1, sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
