from .config import training_config, network_config, env_config
from ..bitboard_env import Connect4Env as Env
from ..network import CNN


def instant_augment(batch):
    raise NotImplementedError("training-side augmentation is outside the self-play path (DESIGN.md section 7)")


def inspect(net, board=None):
    """root priors/values for X on the empty board and O after X plays the centre (Connect4/utils.py:90-111)."""
    e0 = Env()
    p0, v0 = net.predict(e0.current_state())
    e1 = Env()
    e1.step(3)
    p1, v1 = net.predict(e1.current_state())
    return p0.flatten(), float(v0[0, 0]), p1.flatten(), float(v1[0, 0])
