"""ctypes front-end of the CPU oracle (oracle/az_oracle.c, oracle/az_oracle_cnn.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by alphazero_amd/ (the product).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libaz_oracle.so")

C4, TTT, NBTTT = 0, 1, 2
EVAL_UNIFORM, EVAL_HASH, EVAL_CNN, EVAL_INJECT = 0, 1, 2, 3
FLIP_NEVER, FLIP_STREAM = 0, 1
GAME_IDS = {"Connect4": C4, "TTT": TTT, "NBTTT": NBTTT}
CNN_BLOB_FLOATS = 500106


class OrcEnv(C.Structure):
    _fields_ = [("game", C.c_int32), ("turn", C.c_int32), ("curr", C.c_int32), ("pad", C.c_int32),
                ("cell", C.c_int8 * 104)]


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    envp = P(OrcEnv)
    ip, fp, u64p, u8p, i8p = P(C.c_int), P(C.c_float), P(C.c_uint64), P(C.c_uint8), P(C.c_int8)
    vp = C.c_void_p
    sig = {
        "orc_n_actions": (C.c_int, [C.c_int]),
        "orc_env_reset": (None, [envp, C.c_int]),
        "orc_env_from_cells": (None, [envp, C.c_int, i8p, C.c_int]),
        "orc_env_place": (C.c_int, [envp, C.c_int]),
        "orc_env_step": (None, [envp, C.c_int]),
        "orc_env_winner": (C.c_int, [envp]),
        "orc_env_full": (C.c_int, [envp]),
        "orc_env_done": (C.c_int, [envp]),
        "orc_env_valid": (C.c_int, [envp, ip]),
        "orc_state_size": (C.c_int, [C.c_int]),
        "orc_env_state": (None, [envp, fp]),
        "orc_env_flip": (None, [envp]),
        "orc_env_flip_action": (C.c_int, [envp, C.c_int]),
        "orc_nbttt_open": (C.c_int, [envp, C.c_int, C.c_int, C.c_int]),
        "orc_state_words": (C.c_int, [C.c_int]),
        "orc_env_pack": (None, [envp, u64p]),
        "orc_env_unpack": (None, [envp, C.c_int, u64p]),
        "orc_eval_hash": (None, [envp, C.c_uint64, fp, fp]),
        "orc_mcts_new": (vp, [C.c_int, C.c_double, C.c_int, C.c_int, C.c_uint64]),
        "orc_mcts_free": (None, [vp]),
        "orc_mcts_reset_tree": (None, [vp]),
        "orc_mcts_set_net": (None, [vp, vp]),
        "orc_mcts_set_flips": (None, [vp, C.c_int, u8p, C.c_int64]),
        "orc_mcts_set_inject": (None, [vp, fp, fp, C.c_int64]),
        "orc_puct": (C.c_float, [C.c_double, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]),
        "orc_mcts_playout": (None, [vp, envp]),
        "orc_mcts_get_action_visits": (C.c_int, [vp, envp, ip, ip]),
        "orc_mcts_prune_root": (None, [vp, C.c_int]),
        "orc_mcts_root_visits": (C.c_int, [vp]),
        "orc_mcts_root_q": (C.c_float, [vp]),
        "orc_mcts_n_nodes": (C.c_int, [vp]),
        "orc_mcts_mean_levels": (C.c_double, [vp]),
        "orc_mcts_stat_playouts": (C.c_int64, [vp]),
        "orc_mcts_stat_evals": (C.c_int64, [vp]),
        "orc_mcts_children": (C.c_int, [vp, ip, C.c_int, ip, ip, fp, fp]),
        "orc_move_probs": (C.c_int, [ip, ip, C.c_int, C.c_double, C.c_int, fp]),
        "orc_selfplay_game": (C.c_int, [vp, envp, C.c_double, C.c_int, C.c_int, C.c_int, u64p, fp, ip, ip, fp, ip, ip]),
        "orc_pure_new": (vp, [C.c_int, C.c_double, C.c_int, C.c_uint64]),
        "orc_pure_free": (None, [vp]),
        "orc_pure_reset_tree": (None, [vp]),
        "orc_pure_set_flips": (None, [vp, C.c_int, u8p, C.c_int64]),
        "orc_uct": (C.c_double, [C.c_double, C.c_int, C.c_int, C.c_int, C.c_double]),
        "orc_rollout_pick": (C.c_int, [envp, C.c_uint64, C.c_int]),
        "orc_rollout": (C.c_int, [envp, C.c_uint64, P(C.c_int64)]),
        "orc_pure_playout": (None, [vp, envp]),
        "orc_pure_get_action": (C.c_int, [vp, envp, ip, ip, P(C.c_double), ip]),
        "orc_pure_prune_root": (None, [vp, C.c_int]),
        "orc_pure_root_visits": (C.c_int, [vp]),
        "orc_pure_root_q": (C.c_double, [vp]),
        "orc_pure_stat_playouts": (C.c_int64, [vp]),
        "orc_pure_stat_levels": (C.c_int64, [vp]),
        "orc_pure_stat_rollout_plies": (C.c_int64, [vp]),
        "orc_cnn_new": (vp, [fp, C.c_int64]),
        "orc_cnn_free": (None, [vp]),
        "orc_cnn_forward": (None, [vp, fp, fp, fp]),
        "orc_cnn_predict": (None, [vp, fp, fp, fp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class Env:
    """Oracle environment with the reference Env's method names (Environment.py:8-91)."""

    def __init__(self, game, _e=None):
        self.game = GAME_IDS.get(game, game)
        self.e = OrcEnv()
        if _e is not None:
            C.memmove(C.byref(self.e), C.byref(_e), C.sizeof(OrcEnv))
        else:
            lib().orc_env_reset(C.byref(self.e), self.game)

    @classmethod
    def from_cells(cls, game, cells):
        g = GAME_IDS.get(game, game)
        env = cls(g)
        a = np.ascontiguousarray(cells, dtype=np.int8).ravel()
        lib().orc_env_from_cells(C.byref(env.e), g, a.ctypes.data_as(C.POINTER(C.c_int8)), a.size)
        return env

    @classmethod
    def from_packed(cls, game, words):
        g = GAME_IDS.get(game, game)
        env = cls(g)
        w = np.ascontiguousarray(words, dtype=np.uint64)
        lib().orc_env_unpack(C.byref(env.e), g, _u64p(w))
        return env

    @classmethod
    def nbttt_open(cls, board, a1, pick):
        env = cls(NBTTT)
        lib().orc_nbttt_open(C.byref(env.e), board, a1, pick)
        return env

    @property
    def turn(self):
        return self.e.turn

    @turn.setter
    def turn(self, v):
        self.e.turn = int(v)

    @property
    def curr(self):
        return self.e.curr

    @property
    def n_actions(self):
        return lib().orc_n_actions(self.game)

    def cells(self):
        n = {C4: 42, TTT: 9, NBTTT: 100}[self.game]
        return np.array(self.e.cell[:n], dtype=np.int8)

    def copy(self):
        return Env(self.game, self.e)

    def reset(self):
        lib().orc_env_reset(C.byref(self.e), self.game)

    def place(self, a):
        return bool(lib().orc_env_place(C.byref(self.e), int(a)))

    def step(self, a):
        lib().orc_env_step(C.byref(self.e), int(a))

    def winPlayer(self):
        return lib().orc_env_winner(C.byref(self.e))

    check_winner = winPlayer

    def check_full(self):
        return bool(lib().orc_env_full(C.byref(self.e)))

    def done(self):
        return bool(lib().orc_env_done(C.byref(self.e)))

    def valid_move(self):
        out = np.zeros(9, dtype=np.int32)
        n = lib().orc_env_valid(C.byref(self.e), _ip(out))
        return [int(x) for x in out[:n]]

    def valid_mask(self):
        v = set(self.valid_move())
        return [a in v for a in range(self.n_actions)]

    def current_state(self):
        n = lib().orc_state_size(self.game)
        out = np.zeros(n, dtype=np.float32)
        lib().orc_env_state(C.byref(self.e), _fp(out))
        shape = {C4: (1, 3, 6, 7), TTT: (1, 3, 3, 3), NBTTT: (1, 28, 3, 3)}[self.game]
        return out.reshape(shape)

    def flip(self, inplace=False):
        t = self if inplace else self.copy()
        lib().orc_env_flip(C.byref(t.e))
        return t

    def flip_action(self, a):
        return lib().orc_env_flip_action(C.byref(self.e), int(a))

    def packed(self):
        w = np.zeros(lib().orc_state_words(self.game), dtype=np.uint64)
        lib().orc_env_pack(C.byref(self.e), _u64p(w))
        return w


def eval_hash(env, seed):
    A = env.n_actions
    probs = np.zeros(A, dtype=np.float32)
    val = np.zeros(1, dtype=np.float32)
    lib().orc_eval_hash(C.byref(env.e), C.c_uint64(seed), _fp(probs), _fp(val))
    return probs, val[0]


def puct(c_init, c_base, Np, n, Q, prior, noise=None):
    if noise is None:
        noise = prior
    return np.float32(lib().orc_puct(c_init, c_base, Np, n, Q, prior, noise))


class CNN:
    def __init__(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self._blob = blob
        self.h = lib().orc_cnn_new(_fp(blob), blob.size)
        if not self.h:
            raise ValueError("bad weight blob (%d floats, expected %d)" % (blob.size, CNN_BLOB_FLOATS))

    def forward(self, state):
        s = np.ascontiguousarray(state, dtype=np.float32).reshape(-1, 126)
        logp = np.zeros((s.shape[0], 7), dtype=np.float32)
        logv = np.zeros((s.shape[0], 3), dtype=np.float32)
        for i in range(s.shape[0]):
            lib().orc_cnn_forward(self.h, _fp(s[i]), _fp(logp[i]), _fp(logv[i]))
        return logp, logv

    def predict(self, state):
        s = np.ascontiguousarray(state, dtype=np.float32).reshape(-1, 126)
        probs = np.zeros((s.shape[0], 7), dtype=np.float32)
        val = np.zeros((s.shape[0], 1), dtype=np.float32)
        for i in range(s.shape[0]):
            lib().orc_cnn_predict(self.h, _fp(s[i]), _fp(probs[i]), _fp(val[i]))
        return probs, val

    def __del__(self):
        try:
            lib().orc_cnn_free(self.h)
        except Exception:
            pass


class MCTS:
    """Oracle MCTS_AZ (MCTS.py:138-171) in eval() mode (no Dirichlet)."""

    def __init__(self, game, c_init=1.5, n_playout=100, eval_kind=EVAL_UNIFORM, eval_seed=0, net=None):
        self.game = GAME_IDS.get(game, game)
        self.n_actions = lib().orc_n_actions(self.game)
        self.h = lib().orc_mcts_new(self.game, c_init, n_playout, eval_kind, C.c_uint64(eval_seed))
        self._keep = []
        self.net = net
        if net is not None:
            lib().orc_mcts_set_net(self.h, net.h)

    def __del__(self):
        try:
            lib().orc_mcts_free(self.h)
        except Exception:
            pass

    def set_flips(self, bits):
        if bits is None:
            lib().orc_mcts_set_flips(self.h, FLIP_NEVER, None, 0)
            return
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        self._keep.append(b)
        lib().orc_mcts_set_flips(self.h, FLIP_STREAM, b.ctypes.data_as(C.POINTER(C.c_uint8)), b.size)

    def set_inject(self, probs, values):
        p = np.ascontiguousarray(probs, dtype=np.float32)
        v = np.ascontiguousarray(values, dtype=np.float32)
        self._keep += [p, v]
        lib().orc_mcts_set_inject(self.h, _fp(p), _fp(v), v.size)

    def get_action_visits(self, env):
        a = np.zeros(9, dtype=np.int32)
        v = np.zeros(9, dtype=np.int32)
        n = lib().orc_mcts_get_action_visits(self.h, C.byref(env.e), _ip(a), _ip(v))
        return [int(x) for x in a[:n]], [int(x) for x in v[:n]]

    def playout(self, env):
        lib().orc_mcts_playout(self.h, C.byref(env.e))

    def prune_root(self, action):
        lib().orc_mcts_prune_root(self.h, int(action))

    def reset(self):
        lib().orc_mcts_reset_tree(self.h)

    @property
    def root_visits(self):
        return lib().orc_mcts_root_visits(self.h)

    @property
    def root_q(self):
        return np.float32(lib().orc_mcts_root_q(self.h))

    @property
    def n_nodes(self):
        return lib().orc_mcts_n_nodes(self.h)

    @property
    def mean_levels(self):
        return lib().orc_mcts_mean_levels(self.h)

    @property
    def stat_playouts(self):
        return lib().orc_mcts_stat_playouts(self.h)

    @property
    def stat_evals(self):
        return lib().orc_mcts_stat_evals(self.h)

    def children(self, path=()):
        path = np.ascontiguousarray(list(path), dtype=np.int32)
        a = np.zeros(9, dtype=np.int32)
        v = np.zeros(9, dtype=np.int32)
        q = np.zeros(9, dtype=np.float32)
        p = np.zeros(9, dtype=np.float32)
        n = lib().orc_mcts_children(self.h, _ip(path), path.size, _ip(a), _ip(v), _fp(q), _fp(p))
        if n < 0:
            return None
        return a[:n].copy(), v[:n].copy(), q[:n].copy(), p[:n].copy()

    def selfplay_game(self, start_env, temp=1.0, first_n_steps=10, reuse_tree=True, max_plies=96):
        A = self.n_actions
        nw = lib().orc_state_words(self.game)
        states = np.zeros((max_plies, nw), dtype=np.uint64)
        pis = np.zeros((max_plies, A), dtype=np.float32)
        players = np.zeros(max_plies, dtype=np.int32)
        actions = np.zeros(max_plies, dtype=np.int32)
        z = np.zeros(max_plies, dtype=np.float32)
        visits = np.zeros((max_plies, A), dtype=np.int32)
        winner = C.c_int(0)
        n = lib().orc_selfplay_game(self.h, C.byref(start_env.e), float(temp), int(first_n_steps), int(bool(reuse_tree)),
                                    max_plies, _u64p(states), _fp(pis), _ip(players), _ip(actions), _fp(z),
                                    C.byref(winner), _ip(visits))
        return dict(n=n, states=states[:n], pis=pis[:n], players=players[:n], actions=actions[:n], z=z[:n],
                    winner=winner.value, visits=visits[:n])


def move_probs(actions, visits, temp, n_actions):
    a = np.ascontiguousarray(actions, dtype=np.int32)
    v = np.ascontiguousarray(visits, dtype=np.int32)
    pi = np.zeros(n_actions, dtype=np.float32)
    best = lib().orc_move_probs(_ip(a), _ip(v), a.size, float(temp), n_actions, _fp(pi))
    return pi, int(a[best])


class PureMCTS:
    """Oracle MCTS (MCTS.py:90-135: UCT selection + rollout value), the Elo arena's opponent (player.py:71-82)."""

    def __init__(self, game, c_init=4.0, n_playout=2000, seed=0):
        self.game = GAME_IDS.get(game, game)
        self.h = lib().orc_pure_new(self.game, float(c_init), int(n_playout), C.c_uint64(seed))
        self.seed = seed
        self._keep = []

    def __del__(self):
        try:
            lib().orc_pure_free(self.h)
        except Exception:
            pass

    def set_flips(self, bits):
        if bits is None:
            lib().orc_pure_set_flips(self.h, FLIP_NEVER, None, 0)
            return
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        self._keep.append(b)
        lib().orc_pure_set_flips(self.h, FLIP_STREAM, b.ctypes.data_as(C.POINTER(C.c_uint8)), b.size)

    def get_action(self, env):
        a = np.zeros(9, dtype=np.int32)
        v = np.zeros(9, dtype=np.int32)
        q = np.zeros(9, dtype=np.float64)
        n = C.c_int(0)
        best = lib().orc_pure_get_action(self.h, C.byref(env.e), _ip(a), _ip(v), q.ctypes.data_as(C.POINTER(C.c_double)),
                                         C.byref(n))
        k = n.value
        return int(best), [int(x) for x in a[:k]], [int(x) for x in v[:k]], q[:k].copy()

    def prune_root(self, action):
        lib().orc_pure_prune_root(self.h, int(action))

    def reset(self):
        lib().orc_pure_reset_tree(self.h)

    @property
    def root_visits(self):
        return lib().orc_pure_root_visits(self.h)

    @property
    def root_q(self):
        return lib().orc_pure_root_q(self.h)

    @property
    def stat_playouts(self):
        return lib().orc_pure_stat_playouts(self.h)

    @property
    def mean_levels(self):
        return lib().orc_pure_stat_levels(self.h) / max(self.stat_playouts, 1)

    @property
    def mean_rollout_plies(self):
        return lib().orc_pure_stat_rollout_plies(self.h) / max(self.stat_playouts, 1)


def uct(c_init, c_base, Np, n, Q):
    return lib().orc_uct(float(c_init), int(c_base), int(Np), int(n), float(Q))


def rollout(env, seed):
    plies = C.c_int64(0)
    v = lib().orc_rollout(C.byref(env.e), C.c_uint64(seed), C.byref(plies))
    return v, plies.value
