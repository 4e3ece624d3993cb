"""Players with the reference's names and signatures (player.py:12-116).  AlphaZeroPlayer is the hot one."""
import numpy as np

from MCTS import MCTS, MCTS_AZ
from utils import policy_value_fn, softmax


class Player:
    def __init__(self):
        self.win_rate = float("nan")
        self.mcts = None

    def reset_player(self):
        pass

    def train(self):
        raise NotImplementedError

    def eval(self):
        raise NotImplementedError

    def get_action(self, *args, **kwargs):
        raise NotImplementedError


class Human(Player):
    def get_action(self, env, *args, **kwargs):
        return int(input("Your move: ")), None


class NetworkPlayer(Player):
    """raw-policy player (player.py:31-59): one network call per move, no search."""

    def __init__(self, net, deterministic=True):
        super().__init__()
        self.net = net
        self.pv_fn = net
        self.deterministic = deterministic

    def train(self):
        self.net.train()

    def eval(self):
        self.net.eval()

    def get_action(self, env, *args, **kwargs):
        valid = env.valid_move()
        probs = np.asarray(self.net.policy(env.current_state()), dtype=np.float32).flatten()[valid]
        full = np.zeros(self.net.n_actions, dtype=np.float32)
        full[valid] = probs
        probs = probs / probs.sum()
        action = valid[int(np.argmax(probs))] if self.deterministic else int(np.random.choice(valid, p=probs))
        return action, full


class MCTSPlayer(Player):
    """pure Monte-Carlo tree search, the Elo arena's opponent (player.py:71-82, Pipeline.py:130)."""

    def __init__(self, c_puct=4, n_playout=2000, policy=None):
        super().__init__()
        self.mcts = MCTS(policy if policy is not None else policy_value_fn, c_puct, n_playout)

    def train(self):
        pass

    def eval(self):
        pass

    def reset_player(self):
        self.mcts.prune_root(-1)

    def get_action(self, env, *args, **kwargs):
        """returns what the reference returns: the (action, child) pair of MCTS.get_action, which Game.start_play
        unpacks as `action, probs, *_` (game.py:20).  The reference then calls prune_root with that PAIR, which is
        never a key of root.children, so the tree is discarded after every move; so it is here."""
        pair = self.mcts.get_action(env)
        self.mcts.prune_root(-1)
        return pair


class AlphaZeroPlayer(Player):
    def __init__(self, policy_value_fn, c_puct=1.5, n_playout=1000, alpha=None, is_selfplay=0):
        super().__init__()
        self.pv_fn = policy_value_fn
        self.mcts = MCTS_AZ(policy_value_fn, c_puct, n_playout, alpha)
        self.is_selfplay = is_selfplay
        self.n_actions = policy_value_fn.n_actions

    def to(self, device="cpu"):
        self.pv_fn.to(device)

    def train(self):
        self.mcts.train()

    def eval(self):
        self.mcts.eval()

    def reset_player(self):
        self.mcts.prune_root(-1)

    def get_action(self, env, temp=0):
        """player.py:101-116.  pi (the stored training target) is computed by K4 on the device from the visit counts;
        the move itself is drawn with np.random.choice on the host from the same float64 distribution the reference
        samples from, which keeps a seeded np.random stream aligned with the reference's."""
        actions, visits = self.mcts.get_action_visits(env)
        _, pi = self.mcts._eng.choose_moves(temp=float(temp))
        action_probs = pi[0].copy()
        if temp == 0:
            v = np.asarray(visits)
            probs = np.zeros(len(visits), dtype=np.float32)
            probs[v == v.max()] = 1 / int((v == v.max()).sum())
        else:
            probs = softmax(np.log(visits) / max(temp, 1e-8))
        action = np.random.choice(actions, p=probs)
        if self.is_selfplay:
            self.mcts.prune_root(action)
        else:
            self.reset_player()
        return action, action_probs
