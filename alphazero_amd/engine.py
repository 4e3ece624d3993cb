"""SelfPlayEngine -- Python face of one libazamd engine handle (one per GPU / per process).

The batched sibling of the reference's MCTS_AZ + AlphaZeroPlayer + Game (MCTS.py:138-171, player.py:85-116,
game.py:35-58): G games searched in lockstep on the device.  All compute happens in HIP kernels behind the
C ABI (include/azamd.h); this class only marshals numpy buffers.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import (EVAL_CNN, EVAL_HASH, EVAL_HOST, EVAL_ROLLOUT, EVAL_ROLLOUT_HASH, EVAL_UNIFORM, FLIP_NEVER, FLIP_RANDOM, FLIP_STREAM, GAME_IDS,
                   MOVE_FIRST_MAX, MOVE_SAMPLE, AzError, check, lib, ptr)

EVALS = {"uniform": EVAL_UNIFORM, "hash": EVAL_HASH, "cnn": EVAL_CNN, "host": EVAL_HOST, "rollout": EVAL_ROLLOUT,
         "rollout_hash": EVAL_ROLLOUT_HASH}
FLIPS = {"never": FLIP_NEVER, "stream": FLIP_STREAM, "random": FLIP_RANDOM}
MOVES = {"first_max": MOVE_FIRST_MAX, "sample": MOVE_SAMPLE}


class SelfPlayEngine:
    def __init__(self, game="Connect4", n_games=1, n_playout=500, c_puct=1.25, evaluator="uniform", eval_seed=0,
                 flip="never", reuse_tree=True, dirichlet_alpha=None, rng_seed=0, move="first_max", device=0,
                 max_plies=0, nodes_per_game=0, game_id_base=0, n_cohorts=0, game_id_stride=1, eval_cache=0):
        _abi.require_gpu()
        self.game = GAME_IDS.get(game, game)
        self.L = lib()
        self.A = self.L.az_n_actions(self.game)
        self.W = self.L.az_state_words(self.game)
        self.G = int(n_games)
        cfg = _abi.az_config()
        cfg.game, cfg.device, cfg.n_games, cfg.n_playout = self.game, device, self.G, int(n_playout)
        cfg.c_init, cfg.c_base = float(c_puct), 500
        cfg.evaluator, cfg.eval_seed = EVALS.get(evaluator, evaluator), int(eval_seed)
        cfg.flip_mode = FLIPS.get(flip, flip)
        cfg.reuse_tree = 1 if reuse_tree else 0
        cfg.dirichlet_alpha = float(dirichlet_alpha) if dirichlet_alpha else 0.0
        cfg.rng_seed, cfg.move_mode = int(rng_seed), MOVES.get(move, move)
        cfg.max_plies, cfg.nodes_per_game, cfg.game_id_base = int(max_plies), int(nodes_per_game), int(game_id_base)
        cfg.n_cohorts = int(n_cohorts)
        cfg.game_id_stride = int(game_id_stride)
        # evaluation cache of AZ_EVAL_CNN engines (include/azamd.h: az_config.eval_cache_entries): 0 / True = auto size,
        # False / negative = off, n > 0 = that many entries
        cfg.eval_cache_entries = 0 if eval_cache is True else (-1 if eval_cache is False else int(eval_cache))
        self.cfg = cfg
        self.n_playout = int(n_playout)
        h = C.c_void_p()
        check(self.L.az_engine_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self._keep = None
        check(self.L.az_get_config(self.h, C.byref(cfg)))   # resolved values of the 0 = auto fields
        self.n_cohorts, self.nodes_per_game = int(cfg.n_cohorts), int(cfg.nodes_per_game)
        self.eval_cache_entries = int(cfg.eval_cache_entries)   # -1: no cache

    def close(self):
        if getattr(self, "h", None):
            self.L.az_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- roots
    def reset_games(self):
        check(self.L.az_reset_games(self.h))

    def set_roots(self, states):
        s = _abi.as_u64(states, (self.G, self.W))
        check(self.L.az_set_roots(self.h, ptr(s)))

    def get_roots(self):
        s = np.zeros((self.G, self.W), dtype=np.uint64)
        w = np.zeros(self.G, dtype=np.int8)
        d = np.zeros(self.G, dtype=np.uint8)
        check(self.L.az_get_roots(self.h, ptr(s), ptr(w), ptr(d)))
        return s, w, d

    def set_flip_stream(self, bits):
        b = np.ascontiguousarray(bits, dtype=np.uint8).reshape(self.G, -1)
        check(self.L.az_set_flip_stream(self.h, ptr(b), b.shape[1]))

    def set_noise(self, alpha):
        check(self.L.az_set_noise(self.h, float(alpha) if alpha else 0.0))

    # ---- search
    def search(self, n_sims=0, sync=True):
        check(self.L.az_search(self.h, int(n_sims)))
        if sync:
            self.sync()

    def sync(self):
        check(self.L.az_engine_sync(self.h))

    def select(self):
        s = np.zeros((self.G, self.W), dtype=np.uint64)
        f = np.zeros(self.G, dtype=np.uint8)
        check(self.L.az_select(self.h, ptr(s), ptr(f)))
        return s, f

    def backup(self, probs, values):
        p = np.ascontiguousarray(probs, dtype=np.float32).reshape(self.G, self.A)
        v = np.ascontiguousarray(values, dtype=np.float32).reshape(self.G)
        check(self.L.az_backup(self.h, ptr(p), ptr(v)))

    def root_children(self):
        G, A = self.G, self.A
        out = dict(actions=np.zeros((G, A), np.int32), visits=np.zeros((G, A), np.int32), q=np.zeros((G, A), np.float32),
                   prior=np.zeros((G, A), np.float32), n=np.zeros(G, np.int32), root_visits=np.zeros(G, np.int32),
                   root_q=np.zeros(G, np.float32))
        check(self.L.az_get_root_children(self.h, ptr(out["actions"]), ptr(out["visits"]), ptr(out["q"]),
                                          ptr(out["prior"]), ptr(out["n"]), ptr(out["root_visits"]),
                                          ptr(out["root_q"])))
        return out

    def set_root_children(self, actions, visits, n_children):
        """test hook (include/azamd.h: az_set_root_children)"""
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.G, self.A)
        v = np.ascontiguousarray(visits, dtype=np.int32).reshape(self.G, self.A)
        n = np.ascontiguousarray(n_children, dtype=np.int32).reshape(self.G)
        check(self.L.az_set_root_children(self.h, ptr(a), ptr(v), ptr(n)))

    # ---- moves
    def choose_moves(self, temp=0.0, temps=None):
        a = np.zeros(self.G, dtype=np.int32)
        pi = np.zeros((self.G, self.A), dtype=np.float32)
        t = None if temps is None else np.ascontiguousarray(temps, dtype=np.float64)
        check(self.L.az_choose_moves(self.h, ptr(t), float(temp), ptr(a), ptr(pi)))
        return a, pi

    def advance(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.G)
        check(self.L.az_advance(self.h, ptr(a)))

    # ---- self-play
    def selfplay(self, n_games, temp=1.0, first_n_steps=10, max_steps=0):
        r = self.L.az_selfplay(self.h, int(n_games), float(temp), int(first_n_steps), int(max_steps))
        check(int(r))
        return int(r)

    def selfplay_rolling(self, n_games, temp=1.0, first_n_steps=10, max_calls=0):
        """az_selfplay's contract on rolling rounds (no lockstep plies, full leaf batches): include/azamd.h"""
        r = self.L.az_selfplay_rolling(self.h, int(n_games), float(temp), int(first_n_steps), int(max_calls))
        check(int(r))
        return int(r)

    def selfplay_rounds(self, n_rounds=0, temp=1.0, first_n_steps=10, refill=True):
        check(self.L.az_selfplay_rounds(self.h, int(n_rounds), float(temp), int(first_n_steps), 1 if refill else 0))

    def selfplay_step(self, temp=1.0, first_n_steps=10, refill=True):
        check(self.L.az_selfplay_step(self.h, float(temp), int(first_n_steps), 1 if refill else 0))

    def set_game_target(self, n):
        check(self.L.az_set_game_target(self.h, int(n)))

    def set_active(self, mask):
        """park (0) / resume (1) slots between calls (include/azamd.h: az_set_active)"""
        m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.G)
        check(self.L.az_set_active(self.h, ptr(m)))

    def traj_count(self):
        g, p = C.c_int64(0), C.c_int64(0)
        check(self.L.az_traj_count(self.h, C.byref(g), C.byref(p)))
        return g.value, p.value

    def traj_fetch(self, clear=True):
        _, n = self.traj_count()
        out = dict(states=np.zeros((n, self.W), np.uint64), pi=np.zeros((n, self.A), np.float32),
                   z=np.zeros(n, np.float32), player=np.zeros(n, np.int8), action=np.zeros(n, np.int8),
                   done=np.zeros(n, np.uint8), game_id=np.zeros(n, np.uint64))
        got = self.L.az_traj_fetch(self.h, n, ptr(out["states"]), ptr(out["pi"]), ptr(out["z"]), ptr(out["player"]),
                                   ptr(out["action"]), ptr(out["done"]), ptr(out["game_id"]), 1 if clear else 0)
        check(int(got))
        assert got == n
        return out

    @property
    def record_bytes(self):
        return self.L.az_traj_record_bytes(self.game)

    def traj_pack_into(self, dev_ptr, max_plies, clear=True):
        """pack finished plies into a caller-owned DEVICE buffer (e.g. a torch uint8 tensor's data_ptr())."""
        n = self.L.az_traj_pack_dev(self.h, int(max_plies), C.c_void_p(dev_ptr), 1 if clear else 0)
        check(int(n))
        return int(n)

    # ---- network
    def load_net(self, blob, kernel=None):
        """az_net_load; kernel: None (keep the engine's choice), "default", "wino", "split" or "direct" (az_net_set_kernel)"""
        b = np.ascontiguousarray(blob, dtype=np.float32).ravel()
        if kernel is not None:
            check(self.L.az_net_set_kernel(self.h, _abi.K3_KINDS[kernel]))
        check(self.L.az_net_load(self.h, ptr(b), b.size))
        self._keep = b

    @property
    def k3_kernel(self):
        k = self.L.az_net_get_kernel(self.h)
        check(int(k))
        return {1: "wino", 2: "split", 3: "direct"}[int(k)]

    def net_forward(self, states):
        s = _abi.as_u64(states).reshape(-1, 2)
        n = s.shape[0]
        logp, logv = np.zeros((n, 7), np.float32), np.zeros((n, 3), np.float32)
        probs, value = np.zeros((n, 7), np.float32), np.zeros((n, 1), np.float32)
        check(self.L.az_net_forward(self.h, n, ptr(s), ptr(logp), ptr(logv), ptr(probs), ptr(value)))
        return logp, logv, probs, value

    def net_forward_static(self, states):
        """az_net_forward_static: the same forward without any device allocation (at most 4096 positions)"""
        s = _abi.as_u64(states).reshape(-1, 2)
        n = s.shape[0]
        logp, logv = np.zeros((n, 7), np.float32), np.zeros((n, 3), np.float32)
        probs, value = np.zeros((n, 7), np.float32), np.zeros((n, 1), np.float32)
        check(self.L.az_net_forward_static(self.h, n, ptr(s), ptr(logp), ptr(logv), ptr(probs), ptr(value)))
        return logp, logv, probs, value

    def stats(self):
        st = _abi.az_stats()
        check(self.L.az_get_stats(self.h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in st._fields_}

    def set_profiling(self, on=True):
        """on: False/0 off, True/1 every kernel category, 2 the CNN launches only (include/azamd.h)"""
        check(self.L.az_set_profiling(self.h, int(on)))

    def get_profile(self):
        ms = np.zeros(4, np.float64)
        n = np.zeros(4, np.int64)
        check(self.L.az_get_profile(self.h, ptr(ms), ptr(n)))
        busy = np.zeros(4, np.float64)
        check(self.L.az_get_profile_busy(self.h, ptr(busy)))
        names = ("cnn", "select", "backup", "fused")
        return {k: dict(ms=float(ms[i]), launches=int(n[i]), busy_ms=float(busy[i])) for i, k in enumerate(names)}

    def reset_stats(self):
        check(self.L.az_reset_stats(self.h))


# ---- stateless batched environment kernels (K1) -----------------------------------------------------------
def env_step_batch(game, states, actions, device=0):
    _abi.require_gpu()
    g = GAME_IDS.get(game, game)
    L = lib()
    W = L.az_state_words(g)
    s = _abi.as_u64(states).reshape(-1, W)
    n = s.shape[0]
    a = None if actions is None else np.ascontiguousarray(actions, dtype=np.int32).reshape(n)
    out = dict(states=np.zeros((n, W), np.uint64), placed=np.zeros(n, np.uint8), winner=np.zeros(n, np.int8),
               done=np.zeros(n, np.uint8), legal=np.zeros(n, np.uint16))
    check(L.az_env_step_batch(g, device, n, ptr(s), ptr(a), ptr(out["states"]), ptr(out["placed"]), ptr(out["winner"]),
                              ptr(out["done"]), ptr(out["legal"])))
    return out


def env_status_batch(game, states, device=0):
    _abi.require_gpu()
    g = GAME_IDS.get(game, game)
    L = lib()
    W = L.az_state_words(g)
    s = _abi.as_u64(states).reshape(-1, W)
    n = s.shape[0]
    out = dict(winner=np.zeros(n, np.int8), done=np.zeros(n, np.uint8), legal=np.zeros(n, np.uint16),
               full=np.zeros(n, np.uint8))
    check(L.az_env_status_batch(g, device, n, ptr(s), ptr(out["winner"]), ptr(out["done"]), ptr(out["legal"]),
                                ptr(out["full"])))
    return out


def env_planes_batch(game, states, device=0):
    _abi.require_gpu()
    g = GAME_IDS.get(game, game)
    L = lib()
    W = L.az_state_words(g)
    s = _abi.as_u64(states).reshape(-1, W)
    n = s.shape[0]
    shape = {0: (n, 3, 6, 7), 1: (n, 3, 3, 3), 2: (n, 28, 3, 3)}[g]
    planes = np.zeros(shape, dtype=np.float32)
    check(L.az_env_planes_batch(g, device, n, ptr(s), ptr(planes)))
    return planes


def env_flip_batch(game, states, device=0):
    _abi.require_gpu()
    g = GAME_IDS.get(game, game)
    L = lib()
    W, A = L.az_state_words(g), L.az_n_actions(g)
    s = _abi.as_u64(states).reshape(-1, W)
    n = s.shape[0]
    flipped = np.zeros((n, W), np.uint64)
    fa = np.zeros((n, A), np.int8)
    check(L.az_env_flip_batch(g, device, n, ptr(s), ptr(flipped), ptr(fa)))
    return flipped, fa


def puct_batch(c_init, c_base, Np, n, Q, P, device=0):
    _abi.require_gpu()
    Np = np.ascontiguousarray(Np, dtype=np.int32)
    n_ = np.ascontiguousarray(n, dtype=np.int32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    P = np.ascontiguousarray(P, dtype=np.float32)
    out = np.zeros(Np.size, dtype=np.float32)
    check(lib().az_puct_batch(device, float(c_init), int(c_base), Np.size, ptr(Np), ptr(n_), ptr(Q), ptr(P), ptr(out)))
    return out
