"""Shared test helpers (no tests here)."""


class RefShapedBuffer:
    """a buffer with exactly the reference ReplayBuffer's tensors, ring pointer and per-ply store()
    (ReplayBuffer.py:18-36,68-83; the reference file itself cannot travel to the GPU box)."""

    def __init__(self, state_dim, capacity, action_dim, row, col, device):
        import torch
        self.state = torch.full((capacity, state_dim, row, col), torch.nan, dtype=torch.float32, device=device)
        self.prob = torch.full((capacity, action_dim), torch.nan, dtype=torch.float32, device=device)
        self.value = torch.full((capacity, 1), torch.nan, dtype=torch.float32, device=device)
        self.winner = torch.full((capacity, 1), 0, dtype=torch.int32, device=device)
        self.next_state = torch.full_like(self.state, torch.nan)
        self.done = torch.zeros((capacity, 1), dtype=torch.bool, device=device)
        self.device, self.current_capacity, self._ptr = device, capacity, 0

    def store(self, state, prob, value, winner, next_state, done):
        import torch
        idx = self._ptr
        self._ptr = (self._ptr + 1) % self.current_capacity
        self.state[idx] = torch.from_numpy(state).float().to(self.device)
        self.prob[idx] = torch.from_numpy(prob).float().to(self.device)
        self.value[idx] = value
        self.winner[idx] = winner
        self.next_state[idx] = torch.from_numpy(next_state).float().to(self.device)
        self.done[idx] = done
        return idx


# ---- the Connect4 CNN weight blob (include/azamd.h: az_net_load): every float32 tensor of the reference state_dict in
# state_dict order (environments/Connect4/Network.py:31-51).  Names and shapes below are that order.
def _bn(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
            (prefix + ".running_var", (c,))]


BLOB_LAYOUT = [("hidden.0.weight", (128, 3, 3, 3))] + _bn("hidden.1", 128)
for _b in (4, 5, 6):
    BLOB_LAYOUT += [("hidden.%d.conv.0.weight" % _b, (128, 128, 3, 3))] + _bn("hidden.%d.conv.1" % _b, 128)
    BLOB_LAYOUT += [("hidden.%d.conv.4.weight" % _b, (128, 128, 1, 1))] + _bn("hidden.%d.conv.5" % _b, 128)
BLOB_LAYOUT += [("policy_head.0.weight", (3, 128, 1, 1))] + _bn("policy_head.1", 3)
BLOB_LAYOUT += [("policy_head.4.weight", (7, 126)), ("policy_head.4.bias", (7,))]
BLOB_LAYOUT += [("value_head.0.weight", (1, 128, 1, 1))] + _bn("value_head.1", 1)
BLOB_LAYOUT += [("value_head.4.weight", (3, 42)), ("value_head.4.bias", (3,))]


def blob_views(blob):
    """name -> numpy view (reshaped, writable if blob is) into a flat float32 blob"""
    import numpy as np
    out, off = {}, 0
    for name, shape in BLOB_LAYOUT:
        n = int(np.prod(shape))
        out[name] = blob[off:off + n].reshape(shape)
        off += n
    assert off == blob.size == 500106, (off, blob.size)
    return out


def blob_to_state_dict(blob):
    """flat blob -> the torch state_dict the reference's CNN.load_state_dict takes (num_batches_tracked = 0)"""
    import numpy as np
    import torch
    sd = {}
    for name, v in blob_views(np.array(blob, dtype=np.float32)).items():
        sd[name] = torch.from_numpy(v.copy())
        if name.endswith("running_var"):
            sd[name[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
    return sd


STRESS_VARIANTS = ("big_t", "small_t", "big_x", "small_x", "bn_jitter")


def stress_blob(blob, variant):
    """Range-stress variants of a weight blob (tests/golden/make_fixtures_cnn4.py documents them): activations pushed far
    outside / deep inside fp16's range by exact power-of-two rescalings of BatchNorm affine parameters, compensated in the
    weights that consume the activation; `bn_jitter` perturbs the BatchNorm statistics of every layer (seeded)."""
    import numpy as np
    b = np.array(blob, dtype=np.float32).copy()
    v = blob_views(b)
    f = np.float32
    if variant in ("big_t", "small_t"):
        k = f(2.0 ** 14) if variant == "big_t" else f(2.0 ** -18)
        for blk in (4, 5, 6):
            v["hidden.%d.conv.1.weight" % blk] *= k
            v["hidden.%d.conv.1.bias" % blk] *= k
            v["hidden.%d.conv.4.weight" % blk] *= f(1) / k
    elif variant in ("big_x", "small_x"):
        k = f(2.0 ** 15) if variant == "big_x" else f(2.0 ** -20)
        v["hidden.1.weight"] *= k
        v["hidden.1.bias"] *= k
        for blk in (4, 5, 6):
            v["hidden.%d.conv.0.weight" % blk] *= f(1) / k
            v["hidden.%d.conv.5.weight" % blk] *= k
            v["hidden.%d.conv.5.bias" % blk] *= k
        v["policy_head.0.weight"] *= f(1) / k
        v["value_head.0.weight"] *= f(1) / k
    elif variant == "bn_jitter":
        rng = np.random.default_rng(7)
        for name, _ in BLOB_LAYOUT:
            if name.endswith("running_var"):
                p = name[:-len("running_var")]
                c = v[name].shape[0]
                sigma = np.sqrt(v[name] + f(1e-5))
                v[p + "running_mean"] += (rng.normal(0, 0.5, c) * sigma).astype(f)
                v[name] *= np.exp2(rng.uniform(-2, 2, c)).astype(f)
                v[p + "weight"] *= np.exp2(rng.uniform(-1, 1, c)).astype(f)
    else:
        raise ValueError(variant)
    return b
