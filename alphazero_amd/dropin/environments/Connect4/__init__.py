from .config import training_config, network_config, env_config
from ..bitboard_env import Connect4Env as Env
from ..network import CNN
from ..augment import mirror_augment


# column axis of each field of a sampled batch (state, prob, value, winner, next_state, done); None = no board axis
_MIRROR_AXIS = (3, 1, None, None, 3, None)


def instant_augment(batch):
    """Doubles a sampled batch with its left-right mirror image (Connect4/utils.py:114-128): rows [0, B) are the
    batch, rows [B, 2B) the mirrored twins (boards and pi reversed along the column axis, scalars repeated)."""
    return mirror_augment(batch, _MIRROR_AXIS)


def inspect(net, board=None):
    """root priors/values for X on the empty board and O after X plays the centre (Connect4/utils.py:90-111)."""
    e0 = Env()
    p0, v0 = net.predict(e0.current_state())
    e1 = Env()
    e1.step(3)
    p1, v1 = net.predict(e1.current_state())
    return p0.flatten(), float(v0[0, 0]), p1.flatten(), float(v1[0, 0])
