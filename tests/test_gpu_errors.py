"""Error behaviour of the C ABI on the GPU (SURVEY 8b conventions): every misuse is an `int` status + az_last_error()
text -- never an abort, a printf or a GPU fault -- and the engine / the device stay usable afterwards."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def E():
    from alphazero_amd import engine
    return engine


def _raises(E, match):
    return pytest.raises(E.AzError, match=match)


def test_create_rejects_bad_configs(E):
    with _raises(E, "bad game"):
        E.SelfPlayEngine("TTT", n_games=0, n_playout=10)
    with _raises(E, "bad game"):
        E.SelfPlayEngine("TTT", n_games=4, n_playout=0)
    with _raises(E, "bad game"):
        E.SelfPlayEngine(7, n_games=4, n_playout=10)
    with _raises(E, "Connect4 only"):
        E.SelfPlayEngine("TTT", n_games=4, n_playout=10, evaluator="cnn")
    with _raises(E, "ordinal"):
        E.SelfPlayEngine("TTT", n_games=4, n_playout=10, device=64)


def test_wrong_call_order_is_a_state_error(E):
    e = E.SelfPlayEngine("Connect4", n_games=8, n_playout=20, evaluator="cnn")
    with _raises(E, "az_net_load"):
        e.search()
    with _raises(E, "az_net_load"):
        e.selfplay_rounds(1)
    with _raises(E, "az_net_load"):
        e.net_forward(np.zeros((3, 2), np.uint64))
    with _raises(E, "500106"):
        e.load_net(np.zeros(1000, np.float32))          # not the reference state_dict
    e.close()
    e = E.SelfPlayEngine("Connect4", n_games=8, n_playout=20, evaluator="hash", flip="stream")
    with _raises(E, "az_set_flip_stream"):
        e.search()
    e.close()
    e = E.SelfPlayEngine("TTT", n_games=8, n_playout=20, evaluator="host")
    with _raises(E, "az_select"):
        e.search()
    e.close()
    e = E.SelfPlayEngine("Connect4", n_games=8, n_playout=20, evaluator="rollout_hash")
    with _raises(E, "az_search"):
        e.select()
    e.close()
    e = E.SelfPlayEngine("TTT", n_games=8, n_playout=20)
    with _raises(E, "AZ_EVAL_CNN"):
        e.selfplay_rounds(1)
    e.close()


def test_advance_rejects_out_of_range_actions(E):
    e = E.SelfPlayEngine("Connect4", n_games=4, n_playout=30, evaluator="hash")
    e.search()
    before, _, _ = e.get_roots()
    with _raises(E, "outside"):
        e.advance([3, 7, 3, 3])
    after, _, _ = e.get_roots()
    assert np.array_equal(before, after)               # nothing was played
    e.advance([3, -1, 3, 3])                            # -1: that slot stays
    after, _, _ = e.get_roots()
    assert np.array_equal(after[1], before[1]) and not np.array_equal(after[0], before[0])
    e.close()


@pytest.mark.parametrize("game,evaluator", [("TTT", "uniform"), ("Connect4", "hash"), ("Connect4", "host"),
                                            ("Connect4", "rollout_hash")])
def test_node_arena_overflow_is_reported_not_faulted(E, game, evaluator):
    """a node arena too small for the search: expansions that do not fit are dropped (the leaf stays a leaf), the
    error bit surfaces as AZ_ERR_ARENA at the next sync, and nothing is written outside the arena."""
    S = 60
    e = E.SelfPlayEngine(game, n_games=16, n_playout=S, evaluator=evaluator, nodes_per_game=12)
    assert e.nodes_per_game == 12
    with _raises(E, "arena overflow"):
        if evaluator == "host":
            for _ in range(S):
                e.select()
                e.backup(np.full((16, e.A), 1.0 / e.A, np.float32), np.zeros(16, np.float32))
            e.sync()
        else:
            e.search()
    rc = e.root_children()                               # the tree that fits is still a consistent tree
    assert (rc["root_visits"] == S).all() and (rc["visits"].sum(1) == S - 1).all()
    assert (rc["n"] <= e.A).all()
    e.close()
    # the device is healthy: a properly sized engine of the same kind searches normally afterwards
    e = E.SelfPlayEngine(game, n_games=16, n_playout=S, evaluator="uniform" if evaluator == "host" else evaluator)
    e.search()
    rc = e.root_children()
    assert (rc["root_visits"] == S).all() and (rc["visits"].sum(1) == S - 1).all()
    e.close()


def test_selfplay_starts_from_an_empty_trajectory_buffer(E):
    """az_selfplay discards what an earlier call left unfetched: counts restart, nothing is appended or overflows."""
    e = E.SelfPlayEngine("TTT", n_games=32, n_playout=20, evaluator="hash", eval_seed=3)
    n = e.selfplay(32, temp=1.0, first_n_steps=2)
    assert n >= 32 and e.traj_count()[0] == n
    n2 = e.selfplay(32, temp=1.0, first_n_steps=2)
    games, plies = e.traj_count()
    assert games == n2 and n2 >= 32
    tr = e.traj_fetch(clear=True)
    assert tr["done"].sum() == n2 and len(tr["z"]) == plies
    assert e.traj_count() == (0, 0)
    with _raises(E, "az_selfplay: bad argument"):
        e.selfplay(0)
    e.close()
