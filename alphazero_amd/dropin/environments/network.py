"""Inference-side stand-in for the reference's Connect4 CNN (environments/Connect4/Network.py:30-85,
environments/NetworkBase.py:14-24) whose forward is the fused MFMA kernel K3.

Same constructor arguments, `.load/.save`, `.predict`, `.policy`, `.device`, `.n_actions`, `.name()`, `.eval()/.train()`.
Training stays with PyTorch-ROCm (the north star keeps it there): train the reference's own torch module and hand
its state_dict over with `load_state_dict` / `sync_from(module)` after each policy update -- that is the hot reload.
"""
import numpy as np

from alphazero_amd import _abi
from alphazero_amd.engine import SelfPlayEngine

BLOB_FLOATS = 500106  # every float32 tensor of the state_dict, in state_dict order (include/azamd.h: az_net_load)


def state_dict_to_blob(sd):
    """torch state_dict -> flat float32 blob.  num_batches_tracked (int64) entries are skipped."""
    parts = []
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if a.dtype != np.float32:
            raise _abi.AzError("state_dict entry %s has dtype %s, expected float32" % (k, a.dtype))
        parts.append(a.ravel())
    blob = np.concatenate(parts) if parts else np.zeros(0, np.float32)
    if blob.size != BLOB_FLOATS:
        raise _abi.AzError("state_dict holds %d float32 values, the Connect4 CNN has %d" % (blob.size, BLOB_FLOATS))
    return np.ascontiguousarray(blob, dtype=np.float32)


def planes_to_packed(state):
    """(B,3,6,7) network planes (env.current_state, env_cython.pyx:143-149) -> packed Connect4 words (B,2)."""
    s = state.detach().cpu().numpy() if hasattr(state, "detach") else np.asarray(state)
    s = s.reshape(-1, 3, 6, 7)
    out = np.zeros((s.shape[0], 2), dtype=np.uint64)
    # stone bit = col*7 + (5-row)
    shift = np.array([[c * 7 + (5 - r) for c in range(7)] for r in range(6)], dtype=np.uint64)
    one = np.uint64(1)
    for b in range(s.shape[0]):
        x = np.bitwise_or.reduce((one << shift)[s[b, 0] == 1], initial=np.uint64(0))
        o = np.bitwise_or.reduce((one << shift)[s[b, 1] == 1], initial=np.uint64(0))
        if s[b, 2, 0, 0] < 0:
            o |= one << np.uint64(63)
        out[b, 0], out[b, 1] = x, o
    return out


class CNN:
    def __init__(self, lr=0, in_dim=3, h_dim=128, out_dim=7, device="cuda"):
        if (in_dim, h_dim, out_dim) != (3, 128, 7):
            raise _abi.AzError("the fused kernel implements the reference architecture only: in_dim=3, h_dim=128, out_dim=7")
        self.lr = lr
        self.device = device
        self.n_actions = out_dim
        self.blob = None
        self._eng = None

    def name(self):
        return "CNN"

    # the torch module switches (Network.py:67,79); inference here is always eval-mode (BatchNorm folded)
    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def to(self, device="cuda"):
        self.device = device
        return self

    @property
    def device_index(self):
        d = str(self.device)
        return int(d.split(":")[1]) if ":" in d else 0

    # ---- weights
    def load_state_dict(self, sd):
        self._set_blob(state_dict_to_blob(sd))

    def sync_from(self, module):
        """hot reload after a training step: module is the reference's torch CNN (or anything with state_dict())."""
        self.load_state_dict(module.state_dict())

    def load(self, path=None):
        """torch checkpoint (state_dict, as NetworkBase.save writes it) or a raw little-endian float32 blob (*.f32).
        Unlike NetworkBase.load (NetworkBase.py:19-24, which prints and carries on with random weights) a failure raises."""
        if path is None:
            return
        if str(path).endswith(".f32"):
            self._set_blob(np.fromfile(path, dtype="<f4"))
            return
        import torch
        self.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))

    def save(self, path=None):
        if path is not None:
            if self.blob is None:
                raise _abi.AzError("no weights to save")
            self.blob.astype("<f4").tofile(path)

    def _set_blob(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
        if blob.size != BLOB_FLOATS:
            raise _abi.AzError("weight blob has %d floats, expected %d" % (blob.size, BLOB_FLOATS))
        self.blob = blob
        if self._eng is not None:
            self._eng.load_net(blob)
        for fn in list(getattr(self, "_listeners", [])):
            fn(blob)

    def on_reload(self, fn):
        """MCTS_AZ registers here so that engines holding these weights follow a reload."""
        self.__dict__.setdefault("_listeners", []).append(fn)

    def export_flat(self):
        if self.blob is None:
            raise _abi.AzError("CNN has no weights: call load()/load_state_dict() first")
        return self.blob

    # ---- inference (K3)
    def _engine(self):
        if self._eng is None:
            self._eng = SelfPlayEngine("Connect4", n_games=1, n_playout=1, evaluator="cnn", device=self.device_index)
            self._eng.load_net(self.export_flat())
        return self._eng

    def forward(self, state):
        """(log_prob (B,7), value_log_prob (B,3)) -- Network.py:60-64."""
        logp, logv, _, _ = self._engine().net_forward(planes_to_packed(state))
        return logp, logv

    def predict(self, state, draw_factor=0.5):
        """(probs (B,7), value (B,1)) float32 numpy, Network.py:78-85."""
        if draw_factor != 0.5:
            raise _abi.AzError("the kernel implements the reference default draw_factor=0.5")
        _, _, probs, value = self._engine().net_forward(planes_to_packed(state))
        return probs, value

    def policy(self, state):
        _, _, probs, _ = self._engine().net_forward(planes_to_packed(state))
        return probs

    def value(self, state):
        _, logv, _, _ = self._engine().net_forward(planes_to_packed(state))
        return np.exp(logv)

    def train_step(self, dataloader, augment):
        raise NotImplementedError("training is outside the self-play path: train the reference's torch CNN with "
                                  "PyTorch-ROCm and call sync_from(module)")
