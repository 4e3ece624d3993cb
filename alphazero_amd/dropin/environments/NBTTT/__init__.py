from .config import training_config, network_config, env_config
from ..bitboard_env import NBTTTEnv as Env
