"""MCTS / MCTS_AZ with the reference's constructor and methods (MCTS.py:90-171), searched by the K2 (+K3) kernels.

One MCTS_AZ object owns one single-slot engine (G = 1).  `get_action_visits(env)` runs `n_playout` playouts on the
device and returns (actions, visits) of the root children in insertion order, like MCTS.py:166-171.

Evaluator dispatch on `policy_value_fn`:
  * the drop-in CNN (environments/network.py)   -> AZ_EVAL_CNN: select, fused CNN and backup all stay on the device;
  * any other object with .predict/.device/.n_actions (MCTS.py:145-146) -> AZ_EVAL_HOST: per playout the leaf position
    comes back (already mirrored when the playout is flipped), its planes are handed to .predict exactly as the
    reference does (a float tensor on .device) and the (probs, value) pair goes to az_backup.
RNG contract: the mirror bit of every playout is drawn with np.random.rand() < 0.5 on the host, one draw per playout in
playout order (env_cython.pyx:192-199 via MCTS.py:142), so a seeded or patched np.random gives the reference's visit
counts bit for bit in eval() mode.  Dirichlet noise (train mode) comes from the device RNG.
"""
import numpy as np

from alphazero_amd import _abi
from alphazero_amd.engine import SelfPlayEngine, env_planes_batch


class RootView:
    """read-only picture of the root after a search: .n_visits, .Q, .children {action: ChildView} in insertion order."""

    class ChildView:
        def __init__(self, n, q, p):
            self.n_visits, self.Q, self.prior = int(n), np.float32(q), np.float32(p)

    def __init__(self, rc=None):
        self.children = {}
        self.n_visits, self.Q = 0, 0
        if rc is not None:
            k = int(rc["n"][0])
            self.n_visits, self.Q = int(rc["root_visits"][0]), np.float32(rc["root_q"][0])
            for i in range(k):
                self.children[int(rc["actions"][0][i])] = RootView.ChildView(rc["visits"][0][i], rc["q"][0][i],
                                                                             rc["prior"][0][i])

    @property
    def is_leaf(self):
        return not self.children


class MCTS:
    def __init__(self, policy_value_fn, c_init=1.5, n_playout=1000, alpha=None):
        self.policy = policy_value_fn
        self.c_init = c_init
        self.c_base = 500
        self.n_playout = n_playout
        self.alpha = alpha
        self.deterministic = False
        self.reuse_tree = True
        self._eng = None
        self._game = None
        self._root_words = None   # position the engine's tree is rooted at (None: no live tree)

    # ---- MCTS.train()/eval() (MCTS.py:100-106): Dirichlet noise for future expansions on/off
    def train(self):
        self.deterministic = False
        if self._eng is not None:
            self._eng.set_noise(self.alpha)

    def eval(self):
        self.deterministic = True
        if self._eng is not None:
            self._eng.set_noise(None)

    @property
    def root(self):
        if self._eng is None or self._root_words is None:
            return RootView()
        return RootView(self._eng.root_children())

    def _on_device(self):
        return hasattr(self.policy, "export_flat")

    def _engine_for(self, env):
        game = env.GAME
        if self._eng is not None and self._game != game:
            self._eng.close()
            self._eng = None
        if self._eng is None:
            dev = getattr(self.policy, "device_index", 0)
            self._eng = SelfPlayEngine(game, n_games=1, n_playout=self.n_playout, c_puct=self.c_init,
                                       evaluator="cnn" if self._on_device() else "host", flip="stream",
                                       reuse_tree=True, dirichlet_alpha=None if self.deterministic else self.alpha,
                                       rng_seed=int(np.random.randint(1 << 31)) if not self.deterministic else 0,
                                       device=dev)
            self._game = game
            self._root_words = None
            if self._on_device():
                self._eng.load_net(self.policy.export_flat())
                self.policy.on_reload(lambda blob, e=self._eng: e.h and e.load_net(blob))
        return self._eng

    def _root_at(self, env):
        """root the engine's tree at env: keep the live (pruned) tree when it already stands there, else start fresh."""
        w = env.packed()
        if self._root_words is None or not np.array_equal(w, self._root_words):
            self._eng.set_roots(w.reshape(1, -1))
            self._root_words = w

    def prune_root(self, node_index):
        """MCTS.py:130-135: re-root on a child (statistics kept) or, for an unknown action such as -1, start over."""
        if self._eng is None or self._root_words is None:
            return
        rc = self._eng.root_children()
        if node_index in [int(a) for a in rc["actions"][0][:int(rc["n"][0])]]:
            self._eng.advance([int(node_index)])
            self._root_words = self._eng.get_roots()[0][0].copy()
        else:
            self._root_words = None

    def _is_rollout(self):
        return type(self.policy).__name__ == "RolloutPolicy"

    def get_action(self, env):
        """MCTS.py:125-128 with utils.policy_value_fn: n_playout playouts of UCT selection + one rollout each
        (float64 arithmetic, like the reference's Python floats), all on the device; returns the
        (action, child) pair of the most visited root child, first maximum in insertion order."""
        if not self._is_rollout():
            raise NotImplementedError("MCTS.get_action runs the reference's own rollout policy (utils.policy_value_fn) "
                                      "on the device; arbitrary host policies are not supported")
        game = env.GAME
        if self._eng is not None and self._game != game:
            self._eng.close()
            self._eng = None
        if self._eng is None:
            hashed = self.policy.hash_seed is not None
            seed = self.policy.rng_seed if self.policy.rng_seed is not None else int(np.random.randint(1 << 31))
            self._eng = SelfPlayEngine(game, n_games=1, n_playout=self.n_playout, c_puct=self.c_init,
                                       evaluator="rollout_hash" if hashed else "rollout",
                                       eval_seed=self.policy.hash_seed or 0, flip="stream", reuse_tree=True,
                                       rng_seed=seed)
            self._game = game
            self._root_words = None
        eng = self._eng
        self._root_at(env)
        n = int(self.n_playout)
        if game != "NBTTT":   # env.random_flip() draws once per playout (MCTS.py:117)
            bits = np.fromiter((np.random.rand() < 0.5 for _ in range(n)), dtype=np.uint8, count=n)
            eng.set_flip_stream(bits.reshape(1, n))
        eng.search(n)
        root = RootView(eng.root_children())
        best = max(root.children.items(), key=lambda an: an[1].n_visits)
        return best


class MCTS_AZ(MCTS):
    def get_action_visits(self, env):
        assert ((self.alpha is not None) or (self.alpha is None and self.deterministic))
        eng = self._engine_for(env)
        self._root_at(env)
        n = int(self.n_playout)
        if self._game != "NBTTT":
            # one np.random.rand() per playout, in playout order: the reference's random_flip stream
            bits = np.fromiter((np.random.rand() < 0.5 for _ in range(n)), dtype=np.uint8, count=n)
            eng.set_flip_stream(bits.reshape(1, n))
        if self._on_device():
            eng.search(n)
        else:
            self._search_host_eval(eng, n)
        rc = eng.root_children()
        k = int(rc["n"][0])
        actions = tuple(int(a) for a in rc["actions"][0][:k])
        visits = tuple(int(v) for v in rc["visits"][0][:k])
        return actions, visits

    def _search_host_eval(self, eng, n):
        import torch
        A = eng.A
        for _ in range(n):
            leaf, flags = eng.select()
            planes = env_planes_batch(self._game, leaf)
            state = torch.from_numpy(planes).float().to(self.policy.device)   # MCTS.py:145
            probs, value = self.policy.predict(state)                         # MCTS.py:146 (also on terminal leaves)
            probs = np.ascontiguousarray(probs, dtype=np.float32).reshape(1, A)
            value = np.ascontiguousarray(value, dtype=np.float32).reshape(1)
            eng.backup(probs, value)
