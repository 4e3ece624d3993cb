// az_rules.hpp -- bitboard game rules for gfx950 kernels (and for the host-side single-env mirror).
//
// Replaces the reference's array-scanning environments on the hot path:
//   Connect4 : env_cython.pyx:76-141,164-199
//   TTT      : environments/TTT/utils.py:5-45, TTT/env.py:62-77
//   NBTTT    : environments/NBTTT/env.py:29-60, NBTTT/utils.py:136-268 (semantic decisions: DESIGN.md)
//
// Packed state words (public format, documented in include/azamd.h):
//   C4   : w0 = X stones, w1 = O stones | turn flag (bit 63 set <=> O to move, turn = -1).
//          stone bit = col*7 + h, h = 0 bottom row (reference row 5) .. 5 top row (reference row 0);
//          bit col*7+6 is an always-zero sentinel so shifted 4-in-row tests cannot wrap.
//   TTT  : w0 = X bits 0..8 (bit = row*3+col) | O bits 16..24 | turn flag bit 63.
//   NBTTT: w0..w2: sub-board b lives in word b/3 at shift 18*(b%3): 9 X bits then 9 O bits (bit = cell);
//          w3 = current sub-board (bits 0..3) | turn flag bit 63.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AZ_HD __host__ __device__ inline
#else
#define AZ_HD inline
#endif

namespace az {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 TURN_FLAG = 1ull << 63;

AZ_HD int popc64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

// ------------------------------------------------------------------------------------------ Connect4
struct C4 {
    static constexpr int GAME = 0, A = 7, W = 2, LG = 8, MAXD = 44, CELLS = 42, MAX_PLIES = 42;
    static constexpr u64 BOARD = 0x0FDFBF7EFDFBFull;  // 6 low bits of each 7-bit column, 7 columns
    struct State {
        u64 w[2];
    };
    AZ_HD static State initial() { return State{{0ull, 0ull}}; }
    AZ_HD static int turn(const State& s) { return (s.w[1] & TURN_FLAG) ? -1 : 1; }
    AZ_HD static u64 xs(const State& s) { return s.w[0]; }
    AZ_HD static u64 os(const State& s) { return s.w[1] & ~TURN_FLAG; }
    // env_cython.pyx:129-137: lowest empty cell of the column; full column -> false, nothing changes.
    AZ_HD static bool place(State& s, int col) {
        if (col < 0 || col > 6) return false;  // before the shifts: an out-of-range column is no move at all
        u64 occ = xs(s) | os(s);
        u64 colmask = 0x3Full << (7 * col);
        u64 mv = (occ + (1ull << (7 * col))) & colmask;
        if (mv == 0) return false;
        if (s.w[1] & TURN_FLAG) s.w[1] |= mv; else s.w[0] |= mv;
        return true;
    }
    AZ_HD static void step(State& s, int col) {  // :139-141
        if (place(s, col)) s.w[1] ^= TURN_FLAG;
    }
    AZ_HD static bool has4(u64 b) {
        u64 m = b & (b >> 7);
        if (m & (m >> 14)) return true;  // horizontal
        m = b & (b >> 1);
        if (m & (m >> 2)) return true;  // vertical
        m = b & (b >> 8);
        if (m & (m >> 16)) return true;  // diagonal /
        m = b & (b >> 6);
        return (m & (m >> 12)) != 0;  // diagonal backslash
    }
    AZ_HD static int cell(const State& s, int r, int c) {  // reference (row, col) -> +1 / -1 / 0
        int bit = c * 7 + (5 - r);
        return (int)((xs(s) >> bit) & 1) - (int)((os(s) >> bit) & 1);
    }
    // exact scan order of env_cython.pyx:79-115; only reached when BOTH colours hold a line
    // (impossible in play, possible for boards handed in through the batch API).
    AZ_HD static int winner_scan(const State& s) {
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 4; c++) {
                int v = cell(s, r, c);
                if (v && v == cell(s, r, c + 1) && v == cell(s, r, c + 2) && v == cell(s, r, c + 3)) return v;
            }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 7; c++) {
                int v = cell(s, r, c);
                if (v && v == cell(s, r + 1, c) && v == cell(s, r + 2, c) && v == cell(s, r + 3, c)) return v;
            }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) {
                int v = cell(s, r, c);
                if (v && v == cell(s, r + 1, c + 1) && v == cell(s, r + 2, c + 2) && v == cell(s, r + 3, c + 3)) return v;
            }
        for (int r = 3; r < 6; r++)
            for (int c = 0; c < 4; c++) {
                int v = cell(s, r, c);
                if (v && v == cell(s, r - 1, c + 1) && v == cell(s, r - 2, c + 2) && v == cell(s, r - 3, c + 3)) return v;
            }
        return 0;
    }
    AZ_HD static int winner(const State& s) {  // :79-115,164-165
        bool wx = has4(xs(s)), wo = has4(os(s));
        if (wx && wo) return winner_scan(s);
        return wx ? 1 : (wo ? -1 : 0);
    }
    // winner of a position reached by ONE move from a position without a line: only the side that has just moved can
    // have made one (the search only ever steps from non-terminal nodes).  hint: unused here.
    AZ_HD static int move_hint(const State&) { return 0; }
    AZ_HD static int winner_moved(const State& s, int) {
        return (s.w[1] & TURN_FLAG) ? (has4(xs(s)) ? 1 : 0) : (has4(os(s)) ? -1 : 0);
    }
    AZ_HD static bool full(const State& s) { return ((xs(s) | os(s)) & BOARD) == BOARD; }  // :76-77
    AZ_HD static u32 legal(const State& s) {  // :120-124: columns with an empty cell, bit per column
        u64 occ = xs(s) | os(s);
        u32 m = 0;
        for (int c = 0; c < 7; c++)
            if (!((occ >> (7 * c + 5)) & 1)) m |= 1u << c;
        return m;
    }
    AZ_HD static u64 mirror(u64 b) {
        u64 r = 0;
        for (int c = 0; c < 7; c++) r |= ((b >> (7 * c)) & 0x7Full) << (7 * (6 - c));
        return r;
    }
    AZ_HD static State flip(const State& s) {  // :168-172
        State t;
        t.w[0] = mirror(xs(s));
        t.w[1] = mirror(os(s)) | (s.w[1] & TURN_FLAG);
        return t;
    }
    AZ_HD static bool symmetric(const State& s) { return mirror(xs(s)) == xs(s) && mirror(os(s)) == os(s); }
    // :174-190: identity on a mirror-symmetric board, else 6-col.  `sym` = symmetric(unflipped state).
    AZ_HD static int flip_action(bool sym, int a) { return sym ? a : 6 - a; }
    static constexpr bool CAN_FLIP = true;
};

// ------------------------------------------------------------------------------------------ TicTacToe
struct TTT {
    static constexpr int GAME = 1, A = 9, W = 1, LG = 16, MAXD = 11, CELLS = 9, MAX_PLIES = 9;
    struct State {
        u64 w[1];
    };
    AZ_HD static State initial() { return State{{0ull}}; }
    AZ_HD static int turn(const State& s) { return (s.w[0] & TURN_FLAG) ? -1 : 1; }
    AZ_HD static u32 xs(const State& s) { return (u32)(s.w[0] & 0x1FF); }
    AZ_HD static u32 os(const State& s) { return (u32)((s.w[0] >> 16) & 0x1FF); }
    AZ_HD static bool place(State& s, int a) {  // TTT/utils.py:40-45
        if (a < 0 || a > 8 || (((xs(s) | os(s)) >> a) & 1)) return false;
        s.w[0] |= (s.w[0] & TURN_FLAG) ? (1ull << (16 + a)) : (1ull << a);
        return true;
    }
    AZ_HD static void step(State& s, int a) {  // TTT/env.py:46-48
        if (place(s, a)) s.w[0] ^= TURN_FLAG;
    }
    AZ_HD static int winner(const State& s) {  // TTT/utils.py:17-30: row i, col i (i=0..2), diag, anti-diag
        const u32 pat[8] = {0x007, 0x049, 0x038, 0x092, 0x1C0, 0x124, 0x111, 0x054};
        u32 x = xs(s), o = os(s);
        for (int i = 0; i < 8; i++) {
            if ((x & pat[i]) == pat[i]) return 1;
            if ((o & pat[i]) == pat[i]) return -1;
        }
        return 0;
    }
    AZ_HD static bool line9(u32 b) {   // any of the 8 lines fully set, no branches
        const u32 rows = b & (b >> 1) & (b >> 2) & 0x049, cols = b & (b >> 3) & (b >> 6) & 0x007;
        const u32 dg = b & (b >> 4) & (b >> 8) & 1u, ad = (b >> 2) & (b >> 4) & (b >> 6) & 1u;
        return (rows | cols | dg | ad) != 0;
    }
    AZ_HD static int move_hint(const State&) { return 0; }
    AZ_HD static int winner_moved(const State& s, int) {   // see C4::winner_moved
        return (s.w[0] & TURN_FLAG) ? (line9(xs(s)) ? 1 : 0) : (line9(os(s)) ? -1 : 0);
    }
    AZ_HD static bool full(const State& s) { return (xs(s) | os(s)) == 0x1FF; }
    AZ_HD static u32 legal(const State& s) { return ~(xs(s) | os(s)) & 0x1FF; }
    AZ_HD static u32 mirror9(u32 b) { return ((b & 0x049) << 2) | (b & 0x092) | ((b & 0x124) >> 2); }
    AZ_HD static State flip(const State& s) {  // TTT/env.py:62-65 (np.fliplr)
        State t;
        t.w[0] = (u64)mirror9(xs(s)) | ((u64)mirror9(os(s)) << 16) | (s.w[0] & TURN_FLAG);
        return t;
    }
    AZ_HD static bool symmetric(const State&) { return false; }
    AZ_HD static int flip_action(bool, int a) { return (a / 3) * 3 + (2 - a % 3); }  // TTT/env.py:67-69: always
    static constexpr bool CAN_FLIP = true;
};

// ------------------------------------------------------------------------------------------ Nine-board TTT
struct NBTTT {
    static constexpr int GAME = 2, A = 9, W = 4, LG = 16, MAXD = 84, CELLS = 81, MAX_PLIES = 81;
    struct State {
        u64 w[4];
    };
    AZ_HD static State initial() { return State{{0ull, 0ull, 0ull, 0ull}}; }
    AZ_HD static int turn(const State& s) { return (s.w[3] & TURN_FLAG) ? -1 : 1; }
    AZ_HD static int curr(const State& s) { return (int)(s.w[3] & 15); }
    AZ_HD static u32 sub(const State& s, int b) { return (u32)((s.w[b / 3] >> (18 * (b % 3))) & 0x3FFFF); }
    AZ_HD static bool place(State& s, int a) {  // NBTTT/env.py:39-47
        int b = curr(s);
        if (a < 0 || a > 8 || b > 8) return false;  // before sub(): a current-board nibble of 9..15 indexes no word
        u32 sb = sub(s, b);
        if (((sb | (sb >> 9)) >> a) & 1) return false;
        int sh = 18 * (b % 3) + ((s.w[3] & TURN_FLAG) ? 9 : 0) + a;
        s.w[b / 3] |= 1ull << sh;
        s.w[3] = (s.w[3] & ~15ull) | (u64)a;
        return true;
    }
    AZ_HD static void step(State& s, int a) {  // NBTTT/env.py:58-60
        if (place(s, a)) s.w[3] ^= TURN_FLAG;
    }
    // NBTTT/utils.py:178-201: rows (X then O per row), columns, then diagonals X, diagonals O.
    AZ_HD static int winner_sub(u32 sb) {
        u32 x = sb & 0x1FF, o = sb >> 9;
        const u32 rc[6] = {0x007, 0x038, 0x1C0, 0x049, 0x092, 0x124};
        for (int i = 0; i < 6; i++) {
            if ((x & rc[i]) == rc[i]) return 1;
            if ((o & rc[i]) == rc[i]) return -1;
        }
        if ((x & 0x111) == 0x111 || (x & 0x054) == 0x054) return 1;
        if ((o & 0x111) == 0x111 || (o & 0x054) == 0x054) return -1;
        return 0;
    }
    AZ_HD static int winner(const State& s) {  // NBTTT/utils.py:238-245: first sub-board with a line
        for (int b = 0; b < 9; b++) {
            int w = winner_sub(sub(s, b));
            if (w) return w;
        }
        return 0;
    }
    // see C4::winner_moved.  hint = the sub-board the move went into (curr of the position before the move): the only
    // sub-board that can hold a line now, and only the mover's stones can form it
    AZ_HD static int move_hint(const State& before) { return curr(before); }
    AZ_HD static int winner_moved(const State& s, int b) {
        if (b > 8) return 0;
        const u32 sb = sub(s, b);
        return (s.w[3] & TURN_FLAG) ? (TTT::line9(sb & 0x1FF) ? 1 : 0) : (TTT::line9(sb >> 9) ? -1 : 0);
    }
    // build-defined: terminal draw when the CURRENT sub-board is full
    // (NBTTT/dependency/Negamax.c:34, utils.c:88-97), not the reference env's all-81 test.
    // a caller-supplied state whose current-board nibble is 9..15 names no sub-board: no legal move, not full
    AZ_HD static bool full(const State& s) {
        if (curr(s) > 8) return false;
        u32 sb = sub(s, curr(s));
        return ((sb | (sb >> 9)) & 0x1FF) == 0x1FF;
    }
    AZ_HD static u32 legal(const State& s) {
        if (curr(s) > 8) return 0u;
        u32 sb = sub(s, curr(s));
        return ~(sb | (sb >> 9)) & 0x1FF;
    }
    AZ_HD static State flip(const State& s) { return s; }  // no augmentation exists for NBTTT
    AZ_HD static bool symmetric(const State&) { return true; }
    AZ_HD static int flip_action(bool, int a) { return a; }
    static constexpr bool CAN_FLIP = false;
};

// splitmix64 finaliser: counter-based RNG and the synthetic "hashed" evaluator (see az_tree.hip).
AZ_HD u64 mix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace az
