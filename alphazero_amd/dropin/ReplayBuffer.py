"""ReplayBuffer -- the sink of self-play and the source of training batches.

Same constructor, tensors (names, shapes, dtypes, NaN-initialised rows), ring pointer and sampling rules as the
reference's ReplayBuffer (ReplayBuffer.py:10-185), so `TrainPipeline` uses it unchanged, plus what the batched engine
needs:

  * `store_many(records)`: packed ply records (device uint8 [n, record_bytes], include/azamd.h) are expanded by the HIP
    unpack kernel straight into the tensors at the ring pointer -- one launch per contiguous piece instead of one
    Python call and six tensor writes per ply (Pipeline.py:61-62);
  * `len(buffer)` is a counter, not a scan of `value` for NaNs (ReplayBuffer.py:38-39) -- the reference pays that scan
    in every `dataloader`, `sample` and `is_full` call;
  * the index tensors of `dataloader` / `sample_balanced` are built on the device the rows live on.

The random draws are made in the reference's order with the same generators (`torch.randperm` / `torch.randint` on
the default CPU generator, `np.random.randint` for `sample`), so a run seeded like a reference run selects the same
rows: tests/golden/replay.npz, recorded from the reference, pins that.
Not a process-wide singleton (quirk q20: the reference's `_initialized` is never set, so its singleton re-allocates
on every construction anyway).
"""
import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset

_FIELDS = ("state", "prob", "value", "winner", "next_state", "done")


def _as_row(x, device):
    """what store() accepts per field: numpy arrays, tensors or Python scalars"""
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float().to(device)
    return x


def _draw(pool, n):
    """up to n distinct entries of the index tensor `pool`, in random order: one torch.randperm(len(pool))"""
    k = min(int(n), pool.numel())
    return pool[torch.randperm(pool.numel())[:k]]


def _quotas(n, n_classes):
    """n split over the classes as evenly as possible, the first n % n_classes classes get one more"""
    return [n // n_classes + (1 if i < n % n_classes else 0) for i in range(n_classes)]


class ReplayBuffer:
    def __init__(self, state_dim, capacity, action_dim, row, col, replay_ratio=0.1, device="cpu",
                 balance_done_value=True):
        nan = float("nan")
        self.state = torch.full((capacity, state_dim, row, col), nan, dtype=torch.float32, device=device)
        self.prob = torch.full((capacity, action_dim), nan, dtype=torch.float32, device=device)
        self.value = torch.full((capacity, 1), nan, dtype=torch.float32, device=device)
        self.winner = torch.zeros((capacity, 1), dtype=torch.int32, device=device)
        self.next_state = torch.full_like(self.state, nan)
        self.done = torch.zeros((capacity, 1), dtype=torch.bool, device=device)
        self.replay_ratio = replay_ratio
        self.device = device
        self.balance_done_value = balance_done_value
        self.current_capacity = capacity
        self._ptr = 0
        self._rows = 0            # rows holding data = what the reference counts as non-NaN `value` entries

    # ---- size
    def __len__(self):
        return self._rows

    def recount(self):
        """the reference's own definition of len (ReplayBuffer.py:38-39), for callers that wrote the tensors directly"""
        self._rows = int((~self.value.isnan()).sum().item())
        return self._rows

    def is_full(self):
        return len(self) >= len(self.state)

    def double(self):
        """ReplayBuffer.py:41-45: grow the window the ring wraps in, up to the allocated rows"""
        if self.current_capacity < len(self.state):
            self.current_capacity = min(len(self.state), self.current_capacity * 2)
            self._ptr = len(self) % len(self.state)
            assert 0 <= self._ptr < self.current_capacity

    def reset(self):
        for name in ("state", "prob", "value", "next_state"):
            getattr(self, name).fill_(float("nan"))
        self.winner.zero_()
        self.done.zero_()
        self._ptr = 0
        self._rows = 0

    def to(self, device="cpu"):
        for name in _FIELDS:
            setattr(self, name, getattr(self, name).to(device))
        self.device = device

    # ---- writing
    def _wrote(self, lo, n):
        self._rows = max(self._rows, lo + n)
        self._ptr = (lo + n) % self.current_capacity

    def store(self, state, prob, value, winner, next_state, done):
        """one ply (ReplayBuffer.py:68-83); returns the row it went to"""
        i = self._ptr
        self.state[i] = _as_row(state, self.device)
        self.prob[i] = _as_row(prob, self.device)
        self.value[i] = value
        self.winner[i] = winner
        self.next_state[i] = _as_row(next_state, self.device)
        self.done[i] = done
        self._wrote(i, 1)
        return i

    def store_many(self, records, game="Connect4"):
        """n packed ply records (cuda uint8 [n, record_bytes]) -> rows, through the HIP unpack kernel; returns the
        first row.  The tensors must live on the GPU: there is no CPU path for the kernel."""
        from alphazero_amd.collector import store_many
        n, cap = int(records.shape[0]), int(self.current_capacity)
        first = store_many(self, records, game)      # advances self._ptr piece by piece, wrapping at current_capacity
        self._rows = max(self._rows, min(first + n, cap))
        return first

    # ---- reading
    def get(self, indices):
        return tuple(getattr(self, name)[indices] for name in _FIELDS)

    def sample(self, batch_size):
        """ReplayBuffer.py:89-92: uniform with replacement, numpy's global generator"""
        return self.get(torch.from_numpy(np.random.randint(0, len(self), batch_size, dtype=np.int64)))

    def _dataloader_indices(self):
        """ReplayBuffer.py:95-148: at most min(replay_ratio * n, 10000) rows (all of them up to 1000), a fifth of them
        game-ending plies split evenly over the outcomes present, the rest other plies; shuffled."""
        total = len(self)
        budget = min(int(total * self.replay_ratio), 10000, total)
        if total <= 1000:
            budget = total
        if budget <= 0:
            raise ValueError("No available data to sample.")
        done = self.done[:total].reshape(-1)
        outcome = self.winner[:total].reshape(-1)
        ending = done.nonzero(as_tuple=True)[0]
        other = (~done).nonzero(as_tuple=True)[0]
        n_ending = max(1, int(budget * 0.2))
        picks = []
        if ending.numel() > 0:
            if self.balance_done_value:      # winner is int32: the reference's "labels are whole numbers" test holds
                labels = outcome[ending].long()
                classes = torch.unique(labels)
                for cls, quota in zip(classes, _quotas(n_ending, classes.numel())):
                    members = ending[(labels == cls).nonzero(as_tuple=True)[0]]
                    if min(quota, members.numel()) > 0:
                        picks.append(_draw(members, quota))
            else:
                picks.append(_draw(ending, n_ending))
        if other.numel() > 0:
            picks.append(_draw(other, budget - n_ending))
        idx = torch.cat(picks) if picks else ending.new_empty(0)
        if idx.numel() == 0:
            raise ValueError("No available data to sample.")
        return idx[torch.randperm(idx.numel())]

    def dataloader(self, batch_size):
        return DataLoader(TensorDataset(*self.get(self._dataloader_indices())), batch_size=batch_size, shuffle=True)

    def _balanced_indices(self):
        """ReplayBuffer.py:153-182: min(n, replay_ratio * allocated rows, 10000) rows, split evenly over the outcomes
        present, drawn WITH replacement inside each outcome; shuffled."""
        total = len(self)
        budget = min(total, min(int(len(self.state) * self.replay_ratio), 10000))
        outcome = self.winner[:total].reshape(-1).long()
        classes = torch.unique(outcome)
        assert classes.numel() <= 3
        picks = []
        for cls, quota in zip(classes, _quotas(budget, max(classes.numel(), 1))):
            members = (outcome == cls).nonzero(as_tuple=True)[0]
            k = min(members.numel(), quota)
            if k > 0:
                picks.append(members[torch.randint(0, members.numel(), (k,))])
        if not picks:
            raise ValueError("No available data to sample.")
        idx = torch.cat(picks)
        return idx[torch.randperm(idx.numel())]

    def sample_balanced(self, batch_size):
        idx = self._balanced_indices()
        return DataLoader(TensorDataset(*self.get(idx)), batch_size=min(idx.numel(), batch_size), shuffle=True)
