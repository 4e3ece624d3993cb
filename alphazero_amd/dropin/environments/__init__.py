"""Plugin loader with the reference's entry point (environments/__init__.py:3-4): load(name) -> module exposing
Env, CNN (Connect4), training_config, network_config, env_config."""
import importlib


def load(env_name):
    return importlib.import_module("%s.%s" % (__name__, env_name))
