from .config import training_config, network_config, env_config
from ..bitboard_env import TTTEnv as Env
from ..augment import mirror_augment

# (state, prob, value, next_state, done, mask): TTT/utils.py:74-89 reverses the 9-vectors end to end
_MIRROR_AXIS = (3, 1, None, 3, None, 1)


def instant_augment(batch):
    """TTT/utils.py:74-89: the batch followed by its mirrored twins."""
    return mirror_augment(batch, _MIRROR_AXIS)
