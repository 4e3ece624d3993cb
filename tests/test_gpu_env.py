"""K1 parity: batched bitboard env kernels (through the C ABI) vs golden vectors and the CPU oracle.  Bit-exact."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GAMES = ["Connect4", "TTT", "NBTTT"]
NA = {"Connect4": 7, "TTT": 9, "NBTTT": 9}


@pytest.fixture(scope="module")
def eng():
    from alphazero_amd import engine
    return engine


@pytest.fixture(scope="module")
def traces():
    return np.load(os.path.join(G, "env_traces.npz"))


def _trace_positions(traces, game):
    """oracle envs of every fixture row (already verified against the fixture in test_oracle_golden)."""
    starts = traces[game + "_game_start"]
    action = traces[game + "_action"]
    envs = []
    for g in range(len(starts) - 1):
        if game == "NBTTT":
            env = orc.Env.nbttt_open(*[int(x) for x in traces["NBTTT_opening"][g]])
        else:
            env = orc.Env(game)
        for i in range(starts[g], starts[g + 1]):
            if action[i] >= 0:
                env.step(int(action[i]))
            envs.append(env.copy())
    return envs


def _mask_bits(mask_row):
    return int(sum(int(b) << a for a, b in enumerate(mask_row)))


@pytest.mark.parametrize("game", GAMES)
def test_step_status_planes_vs_golden(eng, traces, game):
    envs = _trace_positions(traces, game)
    packed = np.array([e.packed() for e in envs])
    action = traces[game + "_action"]
    n = len(envs)
    # status of every recorded position
    st = eng.env_status_batch(game, packed)
    assert np.array_equal(st["winner"], traces[game + "_winner"].astype(np.int8))
    assert np.array_equal(st["done"].astype(bool), traces[game + "_done"].astype(bool))
    want_legal = np.array([_mask_bits(m) for m in traces[game + "_mask"]], dtype=np.uint16)
    assert np.array_equal(st["legal"], want_legal)
    # planes (env.current_state)
    planes = eng.env_planes_batch(game, packed)
    assert np.array_equal(planes.reshape(n, -1), traces[game + "_state"])
    # transitions: position i --action[i+1]--> position i+1 inside one game
    idx = np.array([i for i in range(n - 1) if action[i + 1] >= 0])
    out = eng.env_step_batch(game, packed[idx], action[idx + 1])
    assert np.array_equal(out["states"], packed[idx + 1])
    assert out["placed"].all()
    assert np.array_equal(out["winner"], traces[game + "_winner"][idx + 1].astype(np.int8))
    assert np.array_equal(out["done"].astype(bool), traces[game + "_done"][idx + 1].astype(bool))
    assert np.array_equal(out["legal"], want_legal[idx + 1])


def test_c4_full_column_and_illegal(eng, traces):
    env = orc.Env.from_cells("Connect4", traces["c4_fullcol_before_cells"])
    env.turn = int(traces["c4_fullcol_before_turn"])
    p = env.packed().reshape(1, 2)
    out = eng.env_step_batch("Connect4", p, [3])
    assert out["placed"][0] == 0 and np.array_equal(out["states"], p)  # silent no-op, turn unchanged (q12)
    out = eng.env_step_batch("TTT", orc.Env("TTT").packed().reshape(1, 1), [4])
    out2 = eng.env_step_batch("TTT", out["states"], [4])
    assert out2["placed"][0] == 0 and np.array_equal(out2["states"], out["states"])


def test_random_boards_scan_order(eng, traces):
    """boards that are not reachable in play (both colours may hold lines): winner follows the reference scan order."""
    envs = [orc.Env.from_cells("Connect4", c) for c in traces["c4_rand_cells"]]
    packed = np.array([e.packed() for e in envs])
    st = eng.env_status_batch("Connect4", packed)
    assert np.array_equal(st["winner"], traces["c4_rand_winner"].astype(np.int8))
    assert np.array_equal(st["full"].astype(bool), traces["c4_rand_full"].astype(bool))
    flipped, fa = eng.env_flip_batch("Connect4", packed)
    assert np.array_equal(fa, traces["c4_rand_flip_action"].astype(np.int8))
    assert np.array_equal(flipped, np.array([e.flip().packed() for e in envs]))
    for game, key, ncell in (("TTT", "ttt", 9), ("NBTTT", "nbttt", 100)):
        es = []
        for c in traces[key + "_rand_cells"]:
            e = orc.Env(game)
            for k in range(ncell):
                e.e.cell[k] = int(c[k])
            es.append(e)
        st = eng.env_status_batch(game, np.array([e.packed() for e in es]))
        assert np.array_equal(st["winner"], traces[key + "_rand_winner"].astype(np.int8))
    fl, fa = eng.env_flip_batch("TTT", np.array([orc.Env("TTT").packed()]))
    assert list(fa[0]) == list(traces["ttt_flip_action"])


@pytest.mark.parametrize("game", GAMES)
def test_random_playouts_vs_oracle_large(eng, game):
    """65 536 concurrent random games driven on the GPU; a sample of them is replayed on the oracle, and
    size-independent invariants are checked on all of them (stone count, turn alternation, done => no moves)."""
    rng = np.random.default_rng(7)
    n = 65536
    A = NA[game]
    if game == "NBTTT":
        start = np.array([orc.Env.nbttt_open(int(b), int(a), int(p)).packed()
                          for b, a, p in rng.integers(0, 9, size=(64, 3))])
        cur = start[rng.integers(0, 64, size=n)]
    else:
        cur = np.tile(orc.Env(game).packed(), (n, 1))
    sample = rng.choice(n, size=48, replace=False)
    oracles = [orc.Env.from_packed(game, cur[i]) for i in sample]
    alive = np.ones(n, dtype=bool)
    st = eng.env_status_batch(game, cur)
    for ply in range(90):
        legal = st["legal"]
        alive = st["done"] == 0
        if not alive.any():
            break
        # pick a random legal action per live game
        r = rng.integers(0, 1 << 30, size=n)
        acts = np.zeros(n, dtype=np.int32)
        for i in np.nonzero(alive)[0][:0]:
            pass
        bits = ((legal[:, None] >> np.arange(A)[None, :]) & 1).astype(bool)
        cnt = bits.sum(1)
        pick = np.where(cnt > 0, r % np.maximum(cnt, 1), 0)
        order = np.cumsum(bits, axis=1) - 1
        acts = np.argmax(bits & (order == pick[:, None]), axis=1).astype(np.int32)
        acts[~alive] = -1 if game != "Connect4" else 7  # illegal on purpose: must be a no-op
        out = eng.env_step_batch(game, cur, acts)
        assert np.array_equal(out["placed"].astype(bool), alive)
        assert np.array_equal(out["states"][~alive], cur[~alive])
        for k, i in enumerate(sample):
            if alive[i]:
                oracles[k].step(int(acts[i]))
            assert np.array_equal(oracles[k].packed(), out["states"][i])
            assert oracles[k].winPlayer() == out["winner"][i]
            assert oracles[k].done() == bool(out["done"][i])
            assert _mask_bits(oracles[k].valid_mask()) == out["legal"][i] or oracles[k].done()
        cur = out["states"]
        st = out
    assert not (st["done"] == 0).any()


@pytest.mark.parametrize("game", GAMES)
def test_step_tiled_path_equals_plain_path(eng, game):
    """the streaming kernel (1024-position tiles, LDS-transposed status stores) and the one-position-per-lane kernel
    (used for tails and small n) must agree byte for byte: 3 tiles + a ragged tail vs the same positions in chunks
    of < 1024."""
    rng = np.random.default_rng(5)
    A = NA[game]
    n = 3 * 1024 + 17
    if game == "NBTTT":
        opens = np.array([orc.Env.nbttt_open(int(b), int(a), int(p)).packed()
                          for b, a, p in rng.integers(0, 9, size=(32, 3))])
        cur = opens[rng.integers(0, 32, size=n)]
    else:
        cur = np.tile(orc.Env(game).packed(), (n, 1))
    for ply in range(7):
        acts = rng.integers(0, A, size=n).astype(np.int32)   # some illegal on purpose
        big = eng.env_step_batch(game, cur, acts)
        for lo in range(0, n, 1000):
            small = eng.env_step_batch(game, cur[lo:lo + 1000], acts[lo:lo + 1000])
            for k in ("states", "placed", "winner", "done", "legal"):
                assert np.array_equal(big[k][lo:lo + 1000], small[k]), (game, ply, k)
        cur = big["states"]
