from ..settings import configs

training_config, network_config, env_config = configs("Connect4")
