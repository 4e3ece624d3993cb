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
