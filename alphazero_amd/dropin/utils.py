"""The helpers of the reference's utils.py that the search path uses (utils.py:26-57)."""
import numpy as np


def softmax(x):
    probs = np.exp(x - np.max(x))
    return probs / np.sum(probs)


class RolloutPolicy:
    """`policy_value_fn` of the pure-MCTS player (utils.py:43-46): uniform priors over the valid moves and the value of
    one random rollout (utils.py:26-40,49-52).  MCTS recognises this object and runs UCT search and rollouts on the
    device (AZ_EVAL_ROLLOUT); it is not called per leaf on the host.
    hash_seed: rollout moves are chosen by a hash of the position instead of the device RNG -- the rule the fixture
    harness substitutes for np.random.rand in rollout_policy_fn, used by the parity tests."""

    def __init__(self, hash_seed=None, rng_seed=None):
        self.hash_seed, self.rng_seed = hash_seed, rng_seed

    def __call__(self, env):
        raise NotImplementedError("the rollout policy runs inside the search kernel (MCTS.get_action); it has no "
                                  "per-leaf host form")


policy_value_fn = RolloutPolicy()
