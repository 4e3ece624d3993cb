"""Abstract environment contract, method for method the one the reference's drivers rely on
(environments/Environment.py:8-91).  The concrete classes in bitboard_env.py satisfy it with the K1 HIP kernels."""
from abc import ABC, abstractmethod


class Environment(ABC):
    @abstractmethod
    def copy(self): ...

    @abstractmethod
    def reset(self): ...

    def done(self):
        return self.check_full() or self.winPlayer() != 0

    @abstractmethod
    def valid_move(self): ...

    @abstractmethod
    def valid_mask(self): ...

    @abstractmethod
    def switch_turn(self): ...

    @abstractmethod
    def place(self, action): ...

    @abstractmethod
    def check_full(self): ...

    @abstractmethod
    def winPlayer(self): ...

    @abstractmethod
    def current_state(self): ...

    @abstractmethod
    def step(self, action): ...

    @abstractmethod
    def show(self): ...
