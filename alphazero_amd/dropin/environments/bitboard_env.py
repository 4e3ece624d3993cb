"""Single-game environments with the reference's Env API, computed by the K1 HIP kernels.

Reference classes mirrored: env_cython.pyx:17-199 (Connect4), environments/TTT/env.py:5-77,
environments/NBTTT/env.py:10-63 (+ the build-defined NBTTT semantics of DESIGN.md section 2).

An Env object is nothing but a packed position (include/azamd.h: "Packed state words").  Every rule question
(place/step, winner, full, legal moves, network planes, mirror, flip_action) is answered by a one-position launch of
the batched kernels through the C ABI -- there is no Python or CPU restatement of the rules here; only format
conversion (float board <-> bit words) happens on the host.  The search never goes through these objects: MCTS_AZ
hands the packed words to the engine and the tree walk steps positions on the device.
"""
import numpy as np

from alphazero_amd import engine as _eng

from .Environment import Environment

_TURN_BIT = np.uint64(1) << np.uint64(63)


def _bit(i):
    return np.uint64(1) << np.uint64(i)


class _BitboardEnv(Environment):
    GAME = None      # "Connect4" | "TTT" | "NBTTT"
    WORDS = 0
    N_ACTIONS = 0

    def __init__(self):
        self._w = np.zeros(self.WORDS, dtype=np.uint64)
        self._st = None

    # ---- packed access (what MCTS_AZ / Game hand to the engine)
    def packed(self):
        return self._w.copy()

    @classmethod
    def from_packed(cls, words):
        e = cls.__new__(cls)
        _BitboardEnv.__init__(e)
        e._w[:] = np.asarray(words, dtype=np.uint64).reshape(cls.WORDS)
        return e

    def _touch(self):
        self._st = None

    def _status(self):
        if self._st is None:
            s = _eng.env_status_batch(self.GAME, self._w.reshape(1, -1))
            self._st = (int(s["winner"][0]), bool(s["done"][0]), int(s["legal"][0]), bool(s["full"][0]))
        return self._st

    # ---- turn
    @property
    def turn(self):
        return -1 if (self._w[-1] & _TURN_BIT) else 1

    @turn.setter
    def turn(self, value):
        if int(value) == -1:
            self._w[-1] |= _TURN_BIT
        else:
            self._w[-1] &= ~_TURN_BIT
        self._touch()

    def switch_turn(self):
        self.turn = -self.turn
        return self.turn

    # ---- rules (device)
    def copy(self):
        return type(self).from_packed(self._w)

    def place(self, action):
        """drop a piece without passing the turn (env_cython.pyx:129-137, TTT/utils.py:40-45)."""
        r = _eng.env_step_batch(self.GAME, self._w.reshape(1, -1), [int(action)])
        placed = bool(r["placed"][0])
        if placed:
            t = self.turn
            self._w[:] = r["states"][0]
            self.turn = t  # the kernel is step = place + switch_turn; undo the switch
        self._touch()
        return placed

    def step(self, action):
        """place, and pass the turn only if the piece was placed (env_cython.pyx:139-141: a full column is a silent
        no-op that keeps the turn)."""
        r = _eng.env_step_batch(self.GAME, self._w.reshape(1, -1), [int(action)])
        self._w[:] = r["states"][0]
        self._st = None

    def check_winner(self):
        return self._status()[0]

    def winPlayer(self):
        return self._status()[0]

    def check_full(self):
        return self._status()[3]

    def done(self):
        return self._status()[1]

    def valid_move(self):
        m = self._status()[2]
        return [a for a in range(self.N_ACTIONS) if (m >> a) & 1]

    def valid_mask(self):
        m = self._status()[2]
        return [bool((m >> a) & 1) for a in range(self.N_ACTIONS)]

    def current_state(self):
        return _eng.env_planes_batch(self.GAME, self._w.reshape(1, -1))

    # ---- mirror augmentation (env_cython.pyx:168-199, TTT/env.py:62-77)
    def flip(self, inplace=False):
        f, _ = _eng.env_flip_batch(self.GAME, self._w.reshape(1, -1))
        target = self if inplace else self.copy()
        target._w[:] = f[0]
        target._touch()
        return target

    def flip_action(self, action):
        _, fa = _eng.env_flip_batch(self.GAME, self._w.reshape(1, -1))
        return int(fa[0][int(action)])

    def random_flip(self, p=0.5):
        env_copy = self.copy()
        flipped = False
        if np.random.rand() < p:  # one draw per call, like the reference: keeps np.random streams aligned
            env_copy.flip(inplace=True)
            flipped = True
        return env_copy, flipped

    def __eq__(self, other):
        return type(other) is type(self) and np.array_equal(self._w, other._w)

    __hash__ = None


class Connect4Env(_BitboardEnv):
    """env_cython.pyx:17-199.  board: float32 (6,7), row 0 = top; +1 = X, -1 = O."""
    GAME, WORDS, N_ACTIONS = "Connect4", 2, 7

    def __init__(self, board=None):
        super().__init__()
        if board is not None:
            self._set_board(np.asarray(board, dtype=np.float32))

    def _set_board(self, b, turn=None):
        x = o = np.uint64(0)
        for r in range(6):
            for c in range(7):
                if b[r, c] == 1:
                    x |= _bit(c * 7 + (5 - r))
                elif b[r, c] == -1:
                    o |= _bit(c * 7 + (5 - r))
        self._w[0], self._w[1] = x, o
        if turn is None:  # env_cython.pyx:52-57: the mover follows from the parity of the stone count
            turn = 1 if int(np.count_nonzero(b)) % 2 == 0 else -1
        self.turn = turn

    @property
    def board(self):
        b = np.zeros((6, 7), dtype=np.float32)
        x, o = int(self._w[0]), int(self._w[1]) & ~(1 << 63)
        for c in range(7):
            for h in range(6):
                k = c * 7 + h
                if (x >> k) & 1:
                    b[5 - h, c] = 1.0
                elif (o >> k) & 1:
                    b[5 - h, c] = -1.0
        return b

    def reset(self):
        self._w[:] = 0
        self._touch()

    def __getstate__(self):  # env_cython.pyx:27-34: (board copy, turn)
        return (self.board, int(self.turn))

    def __setstate__(self, state):
        board, turn = state
        _BitboardEnv.__init__(self)
        self._set_board(np.asarray(board, dtype=np.float32), int(turn))

    def show(self):
        b = self.board
        print("=" * 20)
        for r in range(6):
            print(" ".join("_" if v == 0 else ("X" if v == 1 else "O") for v in b[r]))
        print("0 1 2 3 4 5 6")
        print("=" * 20)


class TTTEnv(_BitboardEnv):
    """environments/TTT/env.py:5-77.  board: float32 (3,3); action = row*3+col."""
    GAME, WORDS, N_ACTIONS = "TTT", 1, 9

    def __init__(self):
        super().__init__()

    @property
    def board(self):
        b = np.zeros((3, 3), dtype=np.float32)
        w = int(self._w[0])
        for i in range(9):
            if (w >> i) & 1:
                b[i // 3, i % 3] = 1.0
            elif (w >> (16 + i)) & 1:
                b[i // 3, i % 3] = -1.0
        return b

    @board.setter
    def board(self, b):
        b = np.asarray(b, dtype=np.float32)
        t = self.turn
        w = np.uint64(0)
        for i in range(9):
            v = b[i // 3, i % 3]
            if v == 1:
                w |= _bit(i)
            elif v == -1:
                w |= _bit(16 + i)
        self._w[0] = w
        self.turn = t

    def reset(self):
        self._w[:] = 0
        self._touch()
        return self.board

    def show(self):
        b = self.board
        print("=" * 10)
        for r in range(3):
            print(" ".join("_" if v == 0 else ("X" if v == 1 else "O") for v in b[r]))
        print("0 1 2\n3 4 5\n6 7 8")
        print("=" * 10)


class NBTTTEnv(_BitboardEnv):
    """environments/NBTTT/env.py:10-63 with the build-defined semantics of DESIGN.md section 2 (the reference class is
    abstract and cannot be instantiated): a game is drawn when the CURRENT sub-board is full, current-board planes are
    indexed 0..8, there is no mirror augmentation.  `boards` is the reference's int (10,10) array (1-based,
    0 = X, 1 = O, 2 = empty); `curr` is the 0-based current sub-board."""
    GAME, WORDS, N_ACTIONS = "NBTTT", 4, 9

    def __init__(self, init_board=None, init_action=None):
        super().__init__()
        # NBTTT/env.py:15-22: two opening plies drawn from np.random, in the reference's order of draws
        curr = int(np.random.randint(9)) if init_board is None else int(init_board)
        action = int(np.random.randint(9)) if init_action is None else int(init_action)
        self._w[3] = np.uint64(curr)
        self.step(action)
        valid = self.valid_move()
        if curr in valid:
            valid.remove(curr)
        self.step(int(np.random.choice(valid)))

    @property
    def curr(self):
        return int(self._w[3]) & 0xF

    @property
    def boards(self):
        b = np.full((10, 10), 2, dtype=np.int64)
        for sb in range(9):
            bits = (int(self._w[sb // 3]) >> (18 * (sb % 3))) & 0x3FFFF
            for cell in range(9):
                if (bits >> cell) & 1:
                    b[sb + 1, cell + 1] = 0
                elif (bits >> (9 + cell)) & 1:
                    b[sb + 1, cell + 1] = 1
        return b

    def reset(self):
        self.__init__()
        return self.boards

    def place(self, action):  # the reference's place(curr, action) is 1-based and private to step(); keep step's view
        return super().place(action)

    def random_flip(self, p=0.5):
        return self.copy(), False

    def flip(self, inplace=False):
        return self if inplace else self.copy()

    def flip_action(self, action):
        return int(action)

    def show(self):
        b = self.boards
        sym = {0: "X", 1: "O", 2: "."}
        for big_r in range(3):
            for r in range(3):
                print(" | ".join(" ".join(sym[int(b[big_r * 3 + bc + 1, r * 3 + c + 1])] for c in range(3))
                                 for bc in range(3)))
            if big_r < 2:
                print("-" * 21)
        print("current sub-board:", self.curr, " to move:", "X" if self.turn == 1 else "O")
