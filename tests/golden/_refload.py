"""Import the reference (read-only, /root/reference) as a live oracle.

CONTAINER-ONLY test infrastructure.  It is used exclusively by
``tests/golden/make_fixtures.py`` to *generate* golden vectors; nothing under
``tests/test_*.py``, ``bench.py`` or ``__graft_entry__`` imports it, because
``/root/reference`` does not exist on the GPU box.

Two shims, both recorded in SURVEY.md section 8(c):
  * ``numba`` is not installed -> an identity ``njit`` stub (the decorated
    functions are plain numpy and behave identically).
  * the reference ships no built ``env_cython`` extension and its tree is
    read-only -> ``oracle/build_ref.sh`` compiles it from the reference's own
    .pyx into ``oracle/_ref/env_cython.so`` and it is registered under the module
    name the reference package expects.
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

REF = os.environ.get("AZ_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_SO = os.path.join(REPO, "oracle", "_ref", "env_cython.so")


def _install_numba_stub():
    if "numba" in sys.modules:
        return
    m = types.ModuleType("numba")

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    m.njit = njit
    m.jit = njit
    sys.modules["numba"] = m


def load_reference():
    """Returns a namespace with the reference's MCTS, player, game, utils and env modules."""
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present: %s" % REF)
    if not os.path.exists(REF_SO):
        raise RuntimeError("run oracle/build_ref.sh first (missing %s)" % REF_SO)
    _install_numba_stub()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True  # never write __pycache__ into the reference
    name = "environments.Connect4.env_cython"
    if name not in sys.modules:
        import environments  # noqa: F401  (reference package root)
        loader = importlib.machinery.ExtensionFileLoader(name, REF_SO)
        spec = importlib.util.spec_from_loader(name, loader)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        loader.exec_module(mod)
    import environments
    ns = types.SimpleNamespace()
    ns.environments = environments
    ns.C4 = environments.load("Connect4")
    ns.TTT = environments.load("TTT")
    import MCTS as ref_mcts
    import player as ref_player
    import game as ref_game
    import utils as ref_utils
    ns.MCTS = ref_mcts
    ns.player = ref_player
    ns.game = ref_game
    ns.utils = ref_utils
    return ns
