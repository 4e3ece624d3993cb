"""Trajectory transport: device trajectory buffer -> packed records -> RCCL all-gather -> ReplayBuffer tensors.

Reference sink: Pipeline.data_collector (Pipeline.py:53-63) calls buffer.store(*data) once per ply with the 6-tuples
of Game.start_self_play (game.py:58); ReplayBuffer (ReplayBuffer.py:18-31,68-83) keeps state/prob/value/winner/
next_state/done tensors with a ring pointer.  Here finished games leave the engine as fixed-size packed records
(48 B per Connect4 ply, include/azamd.h), are all-gathered across the GPUs of the node (torch.distributed "nccl" = RCCL
over xGMI; nothing else crosses GPUs during self-play) and are expanded by a HIP kernel into tensors of exactly the
reference buffer's shapes and dtypes.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi

PLANES = {0: (3, 6, 7), 1: (3, 3, 3), 2: (28, 3, 3)}


def gather_records(local, group=None, force=False):
    """All-gather a ragged [n_i, rec] uint8 tensor across ranks (counts first, then a padded all_gather).
    Pure torch.distributed: works with nccl (device tensors) and gloo (CPU tensors, used by the CPU tests).
    `force` runs the collectives in a group of one as well (tools/rccl_selftest.py)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return local
    world = dist.get_world_size(group)
    rec = local.shape[1]
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    if mx == 0:
        return local
    padded = torch.zeros((mx, rec), dtype=torch.uint8, device=local.device)
    padded[:local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


class TrajectoryGatherer:
    def __init__(self, engine, device, group=None):
        self.eng, self.device, self.group = engine, device, group
        self.rec = engine.record_bytes

    def gather(self):
        _, n = self.eng.traj_count()
        local = torch.empty((n, self.rec), dtype=torch.uint8, device=self.device)
        if n > 0:
            got = self.eng.traj_pack_into(local.data_ptr(), n, clear=True)
            assert got == n
        return gather_records(local, self.group)


class ReplaySink:
    """Device-resident tensors with the reference ReplayBuffer's names, shapes and dtypes (ReplayBuffer.py:20-31),
    filled in bulk from packed records (the `store_many` the reference lacks)."""

    def __init__(self, game, capacity, device):
        self.game = _abi.GAME_IDS.get(game, game)
        L = _abi.lib()
        self.A = L.az_n_actions(self.game)
        self.rec = L.az_traj_record_bytes(self.game)
        self.W = L.az_state_words(self.game)
        c, h, w = PLANES[self.game]
        self.capacity, self.device = int(capacity), device
        self.state = torch.full((capacity, c, h, w), float("nan"), dtype=torch.float32, device=device)
        self.prob = torch.full((capacity, self.A), float("nan"), dtype=torch.float32, device=device)
        self.value = torch.full((capacity, 1), float("nan"), dtype=torch.float32, device=device)
        self.winner = torch.zeros((capacity, 1), dtype=torch.int32, device=device)
        self.next_state = torch.full_like(self.state, float("nan"))
        self.done = torch.zeros((capacity, 1), dtype=torch.bool, device=device)
        self._ptr = 0
        self.total_stored = 0
        self.games_seen = 0
        self.plies_seen = 0

    def __len__(self):
        return min(self.total_stored, self.capacity)

    def _unpack_into(self, recs, lo, n):
        L = _abi.lib()
        vp = C.c_void_p
        sl = slice(lo, lo + n)
        done_u8 = self.done[sl].view(torch.uint8)
        _abi.check(L.az_traj_unpack_dev(self.game, n, vp(recs.data_ptr()), vp(self.state[sl].data_ptr()),
                                        vp(self.prob[sl].data_ptr()), vp(self.value[sl].data_ptr()),
                                        vp(self.winner[sl].data_ptr()), vp(self.next_state[sl].data_ptr()),
                                        vp(done_u8.data_ptr())))

    def store_packed(self, recs):
        n = int(recs.shape[0])
        if n == 0:
            return 0
        recs = recs.contiguous()
        if recs.device.type != "cuda":
            raise _abi.AzError("ReplaySink.store_packed needs device-resident records (no CPU path)")
        done_col = self.W * 8 + self.A * 4 + 3
        self.games_seen += int(recs[:, done_col].sum().item())
        self.plies_seen += n
        off = 0
        while off < n:  # ring buffer: at most two contiguous pieces per lap
            m = min(n - off, self.capacity - self._ptr)
            self._unpack_into(recs[off:off + m], self._ptr, m)
            self._ptr = (self._ptr + m) % self.capacity
            off += m
        self.total_stored += n
        return n

    def store_into(self, buffer):
        """copy everything held here into a reference-style ReplayBuffer object (ReplayBuffer.py:68-83 semantics)."""
        n = len(self)
        for i in range(n):
            buffer.store(self.state[i], self.prob[i], self.value[i], self.winner[i], self.next_state[i], self.done[i])
        return n


def store_many(buffer, recs, game="Connect4"):
    """The bulk form of `for data in play_data: buffer.store(*data)` (Pipeline.py:61-62) on the REFERENCE's own
    ReplayBuffer object (ReplayBuffer.py:10-83): n packed records (device uint8 [n, record_bytes], e.g. from
    TrajectoryGatherer.gather) are expanded by the unpack kernel straight into buffer.state / prob / value / winner /
    next_state / done at buffer._ptr, with store()'s ring semantics (wrap at buffer.current_capacity) -- one kernel
    launch per contiguous piece instead of one Python call and six tensor writes per ply.
    The buffer's tensors must live on the GPU (ReplayBuffer(..., device='cuda')); returns the index of the first row."""
    g = _abi.GAME_IDS.get(game, game)
    L = _abi.lib()
    n = int(recs.shape[0])
    first = int(buffer._ptr)
    if n == 0:
        return first
    recs = recs.contiguous()
    for t in (buffer.state, buffer.prob, buffer.value, buffer.winner, buffer.next_state, buffer.done):
        if t.device.type != "cuda" or not t.is_contiguous():
            raise _abi.AzError("store_many needs contiguous device-resident ReplayBuffer tensors (no CPU path)")
    if recs.device.type != "cuda" or recs.dtype != torch.uint8 or recs.shape[1] != L.az_traj_record_bytes(g):
        raise _abi.AzError("store_many: records must be a cuda uint8 [n, %d] tensor" % L.az_traj_record_bytes(g))
    if tuple(buffer.state.shape[1:]) != PLANES[g] or buffer.prob.shape[1] != L.az_n_actions(g):
        raise _abi.AzError("store_many: buffer shapes do not match game %r" % (game,))
    cap = int(buffer.current_capacity)
    vp = C.c_void_p
    off = 0
    while off < n:
        lo = int(buffer._ptr)
        m = min(n - off, cap - lo)
        sl = slice(lo, lo + m)
        _abi.check(L.az_traj_unpack_dev(g, m, vp(recs[off:off + m].data_ptr()), vp(buffer.state[sl].data_ptr()),
                                        vp(buffer.prob[sl].data_ptr()), vp(buffer.value[sl].data_ptr()),
                                        vp(buffer.winner[sl].data_ptr()), vp(buffer.next_state[sl].data_ptr()),
                                        vp(buffer.done[sl].view(torch.uint8).data_ptr())))
        buffer._ptr = (lo + m) % cap
        off += m
    return first
