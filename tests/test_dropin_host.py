"""Host-side logic of the drop-in mirrors that needs no GPU: plugin loader, configs, format conversions, weight blob
export.  (Everything that computes runs HIP kernels and is covered by tests/test_gpu_dropin.py.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "alphazero_amd", "dropin")
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def dropin():
    sys.path.insert(0, DROPIN)
    import environments
    yield environments
    sys.path.remove(DROPIN)


def test_load_exposes_reference_attributes(dropin):
    """environments/__init__.py:3-4 + Connect4/__init__.py:1-4: load(name) -> module with Env, CNN, configs."""
    m = dropin.load("Connect4")
    for name in ("Env", "CNN", "training_config", "network_config", "env_config", "config", "inspect"):
        assert hasattr(m, name), name
    # values on the self-play path (Connect4/config.py:1-21)
    t = m.config.training_config
    assert (t["c_puct"], t["n_playout"], t["dirichlet_alpha"], t["temp"], t["first_n_steps"]) == (1.25, 80, 0.7, 1.0, 10)
    assert dropin.load("TTT").training_config["c_puct"] == 1.0
    assert dropin.load("NBTTT").training_config["first_n_steps"] == 12
    assert dropin.load("TTT").Env.N_ACTIONS == 9 and dropin.load("NBTTT").Env.WORDS == 4


def test_planes_to_packed_roundtrip_on_golden_states(dropin):
    """(B,3,6,7) planes recorded from the reference -> packed words -> cells: same stones, same mover."""
    from environments.network import planes_to_packed
    tr = np.load(os.path.join(G, "env_traces.npz"))
    planes = tr["Connect4_state"][:400].reshape(-1, 3, 6, 7)
    cells = tr["Connect4_cells"][:400].reshape(-1, 6, 7)
    turn = tr["Connect4_turn"][:400]
    w = planes_to_packed(planes)
    for i in range(len(w)):
        x, o = int(w[i, 0]), int(w[i, 1])
        assert (o >> 63) == (1 if turn[i] == -1 else 0)
        o &= ~(1 << 63)
        for r in range(6):
            for c in range(7):
                bit = c * 7 + (5 - r)
                assert ((x >> bit) & 1) == (cells[i, r, c] == 1)
                assert ((o >> bit) & 1) == (cells[i, r, c] == -1)


def test_state_dict_to_blob_layout(dropin):
    """blob = every float32 tensor of the state_dict in order, num_batches_tracked skipped; wrong sizes raise."""
    import torch
    from alphazero_amd._abi import AzError
    from environments.network import BLOB_FLOATS, CNN, state_dict_to_blob
    blob = np.fromfile(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    assert blob.size == BLOB_FLOATS
    # rebuild a state_dict with the reference's tensor shapes (Connect4/Network.py:31-51) from the blob
    shapes = [("hidden.0.weight", (128, 3, 3, 3))] + _bn("hidden.1", 128)
    for b in (4, 5, 6):
        shapes += [("hidden.%d.conv.0.weight" % b, (128, 128, 3, 3))] + _bn("hidden.%d.conv.1" % b, 128)
        shapes += [("hidden.%d.conv.4.weight" % b, (128, 128, 1, 1))] + _bn("hidden.%d.conv.5" % b, 128)
    shapes += [("policy_head.0.weight", (3, 128, 1, 1))] + _bn("policy_head.1", 3)
    shapes += [("policy_head.4.weight", (7, 126)), ("policy_head.4.bias", (7,))]
    shapes += [("value_head.0.weight", (1, 128, 1, 1))] + _bn("value_head.1", 1)
    shapes += [("value_head.4.weight", (3, 42)), ("value_head.4.bias", (3,))]
    sd, off = {}, 0
    for k, shp in shapes:
        if shp is None:
            sd[k] = torch.tensor(7, dtype=torch.int64)
            continue
        n = int(np.prod(shp))
        sd[k] = torch.from_numpy(blob[off:off + n].reshape(shp).copy())
        off += n
    assert off == BLOB_FLOATS
    assert np.array_equal(state_dict_to_blob(sd), blob)
    del sd["value_head.4.bias"]
    with pytest.raises(AzError):
        state_dict_to_blob(sd)
    net = CNN(0, device="cuda")
    with pytest.raises(AzError):
        net.export_flat()            # no weights yet: loud, unlike NetworkBase.load (q14)
    with pytest.raises(AzError):
        CNN(0, h_dim=64)


def test_instant_augment_mirrors_batch(dropin):
    """Connect4/utils.py:114-128 and TTT/utils.py:74-89 restated in numpy: originals, then their column mirrors."""
    import torch
    rng = np.random.default_rng(5)
    B = 6
    c4 = (rng.random((B, 3, 6, 7), dtype=np.float32), rng.random((B, 7), dtype=np.float32),
          rng.random((B, 1), dtype=np.float32), rng.integers(-1, 2, (B, 1)).astype(np.int32),
          rng.random((B, 3, 6, 7), dtype=np.float32), rng.integers(0, 2, (B, 1)).astype(bool))
    out = dropin.load("Connect4").instant_augment(tuple(torch.from_numpy(x) for x in c4))
    axes = (3, 1, None, None, 3, None)
    assert len(out) == 6
    for o, x, ax in zip(out, c4, axes):
        want = np.concatenate([x, x if ax is None else np.flip(x, ax)], 0)
        assert o.dtype == torch.from_numpy(x).dtype and np.array_equal(o.numpy(), want)
    # the mirror of a position's pi belongs to the mirrored position: column c <-> 6 - c
    assert np.array_equal(out[1][B:, 0].numpy(), c4[1][:, 6]) and np.array_equal(out[0][B:, :, :, 0].numpy(), c4[0][:, :, :, 6])
    ttt = (rng.random((B, 3, 3, 3), dtype=np.float32), rng.random((B, 9), dtype=np.float32),
           rng.random((B, 1), dtype=np.float32), rng.random((B, 3, 3, 3), dtype=np.float32),
           rng.integers(0, 2, (B, 1)).astype(bool), rng.integers(0, 2, (B, 9)).astype(bool))
    out = dropin.load("TTT").instant_augment(tuple(torch.from_numpy(x) for x in ttt))
    for o, x, ax in zip(out, ttt, (3, 1, None, 3, None, 1)):
        assert np.array_equal(o.numpy(), np.concatenate([x, x if ax is None else np.flip(x, ax)], 0))
    with pytest.raises(ValueError):
        dropin.load("TTT").instant_augment(tuple(torch.from_numpy(x) for x in c4[:5]))


def test_instant_augment_vs_reference_fixture(dropin):
    """the drop-in's instant_augment against tensors recorded from the reference's own functions (Connect4/utils.py:114-128,
    TTT/utils.py:74-89; tests/golden/augment.npz, made by make_fixtures_augment.py): same values, shapes and dtypes."""
    import torch
    d = np.load(os.path.join(G, "augment.npz"))
    for game, key in (("Connect4", "c4"), ("TTT", "ttt")):
        aug = dropin.load(game).instant_augment
        for B in (1, 5, 64):
            batch = tuple(torch.from_numpy(d["%s/%d/in%d" % (key, B, i)]) for i in range(6))
            out = aug(batch)
            assert len(out) == 6
            for i, o in enumerate(out):
                want = d["%s/%d/out%d" % (key, B, i)]
                assert o.dtype == torch.from_numpy(want).dtype and tuple(o.shape) == want.shape, (game, B, i)
                assert np.array_equal(o.numpy(), want), (game, B, i)


def test_replaybuffer_hands_out_the_reference_rows(dropin):
    """dropin/ReplayBuffer.py vs rows recorded from the reference's ReplayBuffer (tests/golden/replay.npz, made by
    make_fixtures_replay.py): same plies stored, same torch / numpy seeds -> the same rows in the same order from
    dataloader (ReplayBuffer.py:94-151), sample_balanced (:153-185) and sample (:89-92); len, ring pointer, is_full."""
    import json
    import torch
    from ReplayBuffer import ReplayBuffer
    sys.path.insert(0, G)
    try:
        from make_fixtures_replay import store_all, synthetic_plies
    finally:
        sys.path.remove(G)
    fx = np.load(os.path.join(G, "replay.npz"))
    meta = json.loads(bytes(fx["meta"]).decode())
    assert len(meta["cases"]) >= 7
    for i, m in enumerate(meta["cases"]):
        done, winner = synthetic_plies(m["n"], m["p_done"], m["outcomes"], 1000 + m["seed"])
        buf = ReplayBuffer(1, m["capacity"], 2, 1, 2, replay_ratio=m["replay_ratio"], device="cpu",
                           balance_done_value=m["balance"])
        assert len(buf) == 0 and not buf.is_full()
        store_all(buf, m["n"], done, winner)
        assert (len(buf), buf._ptr, buf.is_full()) == (m["len"], m["ptr"], m["is_full"]), i
        assert buf.recount() == m["len"]                       # the counter agrees with the reference's NaN scan
        torch.manual_seed(m["seed"])
        dl = buf.dataloader(32)
        got = dl.dataset.tensors[2].reshape(-1).numpy().astype(np.int64)
        assert np.array_equal(got, fx["case%d_dataloader" % i]), i
        batch = next(iter(dl))
        assert len(batch) == 6 and batch[0].shape[1:] == (1, 1, 2) and batch[5].dtype == torch.bool
        torch.manual_seed(m["seed"] + 100)
        dl = buf.sample_balanced(32)
        assert np.array_equal(dl.dataset.tensors[2].reshape(-1).numpy().astype(np.int64), fx["case%d_balanced" % i]), i
        assert dl.batch_size == m["balanced_batch"]
        np.random.seed(m["seed"])
        assert np.array_equal(buf.sample(50)[2].reshape(-1).numpy().astype(np.int64), fx["case%d_sample" % i]), i
    # rows keep what store() was given (value k at the row the ring pointer named), reset() empties the buffer
    k = int(buf.value[7, 0])
    assert buf.state[7, 0, 0, 0] == k and buf.next_state[7, 0, 0, 1] == k + 0.5 and buf.prob[7, 1] == -k
    buf.reset()
    assert len(buf) == 0 and buf._ptr == 0 and bool(buf.value.isnan().all())
    with pytest.raises(ValueError):
        buf.dataloader(8)
    # double(): the ring window grows up to the allocated rows and the pointer moves behind the data
    buf.current_capacity = 500
    store_all(buf, 500, done, winner)
    assert (len(buf), buf._ptr) == (500, 0)
    buf.double()
    assert (buf.current_capacity, buf._ptr) == (1000, 500)
    with pytest.raises(Exception):
        buf.store_many(torch.zeros((4, 48), dtype=torch.uint8))   # CPU tensors: the unpack kernel has no CPU path


def _bn(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
            (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", None)]
