/*
 * az_oracle.c -- CPU restatement of the reference self-play hot path.
 *
 *   TEST INFRASTRUCTURE ONLY.  Nothing in alphazero_amd/ (the product) may include,
 *   link, import or call this file.  Only tests/, __graft_entry__.smoke() and
 *   bench.py's cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
 * against golden vectors produced by running the reference itself in the build
 * container (tests/golden/make_fixtures.py imports /root/reference; the reference
 * ships no tests or fixtures of its own).
 *
 * The oracle deliberately keeps the reference's data representation (cell arrays,
 * window scans, pointer-free but insertion-ordered child lists) so that it is an
 * independent statement of the algorithm from the bitboard / SoA HIP kernels.
 *
 * Reference citations are relative to /root/reference.
 *
 * Build: gcc -O2 -ffp-contract=off (tree/PUCT arithmetic must not be FMA-contracted;
 * the CNN part is compiled separately in az_oracle_cnn.c where contraction is harmless).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "az_oracle.h"

/* ------------------------------------------------------------------ helpers */

int orc_n_actions(int game) { return game == ORC_C4 ? 7 : 9; }

static uint64_t mix64(uint64_t z) { /* splitmix64 finaliser (public domain constants) */
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* ------------------------------------------------------------------ env    */

/* env_cython.pyx:42-57,65-67 ; TTT/env.py:6-9 ; NBTTT/env.py:13-14 (opening handled by orc_nbttt_open) */
void orc_env_reset(orc_env *e, int game) {
    memset(e, 0, sizeof(*e));
    e->game = game;
    e->turn = 1;
    e->curr = 0;
    if (game == ORC_NBTTT)
        for (int i = 0; i < 100; i++) e->cell[i] = 2;
}

/* env_cython.pyx:42-57: Env(board) infers the turn from piece-count parity. */
void orc_env_from_cells(orc_env *e, int game, const int8_t *cells, int n) {
    orc_env_reset(e, game);
    int count = 0;
    for (int i = 0; i < n; i++) {
        e->cell[i] = cells[i];
        if (cells[i] != 0) count++;
    }
    e->turn = (count % 2 == 0) ? 1 : -1;
}

/* env_cython.pyx:129-137 ; TTT/utils.py:40-45 ; NBTTT/env.py:39-47 */
int orc_env_place(orc_env *e, int action) {
    if (e->game == ORC_C4) {
        for (int row = 5; row >= 0; row--) {
            if (e->cell[row * 7 + action] == 0) {
                e->cell[row * 7 + action] = (int8_t)e->turn;
                return 1;
            }
        }
        return 0;
    } else if (e->game == ORC_TTT) {
        if (e->cell[action] == 0) {
            e->cell[action] = (int8_t)e->turn;
            return 1;
        }
        return 0;
    } else {
        int r = e->curr + 1, c = action + 1; /* NBTTT/env.py:58-60 passes curr+1, action+1 */
        if (e->cell[r * 10 + c] == 2) {
            e->cell[r * 10 + c] = (e->turn == 1) ? 0 : 1;
            e->curr = c - 1;
            return 1;
        }
        return 0;
    }
}

/* env_cython.pyx:126-127,139-141 ; TTT/env.py:30-32,46-48 ; NBTTT/env.py:35-37,58-60 */
void orc_env_step(orc_env *e, int action) {
    if (orc_env_place(e, action)) e->turn = -e->turn;
}

static int c4_winner(const int8_t *b) { /* env_cython.pyx:79-115, same scan order */
    int r, c, cur;
    for (r = 0; r < 6; r++)
        for (c = 0; c < 4; c++) {
            cur = b[r * 7 + c];
            if (cur != 0 && cur == b[r * 7 + c + 1] && cur == b[r * 7 + c + 2] && cur == b[r * 7 + c + 3]) return cur;
        }
    for (r = 0; r < 3; r++)
        for (c = 0; c < 7; c++) {
            cur = b[r * 7 + c];
            if (cur != 0 && cur == b[(r + 1) * 7 + c] && cur == b[(r + 2) * 7 + c] && cur == b[(r + 3) * 7 + c]) return cur;
        }
    for (r = 0; r < 3; r++)
        for (c = 0; c < 4; c++) {
            cur = b[r * 7 + c];
            if (cur != 0 && cur == b[(r + 1) * 7 + c + 1] && cur == b[(r + 2) * 7 + c + 2] && cur == b[(r + 3) * 7 + c + 3]) return cur;
        }
    for (r = 3; r < 6; r++)
        for (c = 0; c < 4; c++) {
            cur = b[r * 7 + c];
            if (cur != 0 && cur == b[(r - 1) * 7 + c + 1] && cur == b[(r - 2) * 7 + c + 2] && cur == b[(r - 3) * 7 + c + 3]) return cur;
        }
    return 0;
}

static int ttt_winner(const int8_t *b) { /* TTT/utils.py:17-30: row i then column i, then diagonals */
    for (int i = 0; i < 3; i++) {
        if (b[i * 3] != 0 && b[i * 3 + 1] == b[i * 3] && b[i * 3 + 2] == b[i * 3]) return b[i * 3];
        if (b[i] != 0 && b[3 + i] == b[i] && b[6 + i] == b[i]) return b[i];
    }
    if (b[0] != 0 && b[0] == b[4] && b[4] == b[8]) return b[0];
    if (b[2] != 0 && b[2] == b[4] && b[4] == b[6]) return b[2];
    return 0;
}

/* NBTTT/utils.py:178-201 on the (X=+1,O=-1) view built by get_board(view=0) :215-235 */
static int nbttt_single(const int8_t *row9 /* 9 cells, 2/0/1 */) {
    int v[9];
    for (int i = 0; i < 9; i++) v[i] = row9[i] == 0 ? 1 : (row9[i] == 1 ? -1 : 0);
    for (int r = 0; r < 3; r++) {
        int s = v[r * 3] + v[r * 3 + 1] + v[r * 3 + 2];
        if (s == 3) return 1;
        if (s == -3) return -1;
    }
    for (int c = 0; c < 3; c++) {
        int s = v[c] + v[3 + c] + v[6 + c];
        if (s == 3) return 1;
        if (s == -3) return -1;
    }
    int d1 = v[0] + v[4] + v[8], d2 = v[2] + v[4] + v[6];
    if (d1 == 3 || d2 == 3) return 1;
    if (d1 == -3 || d2 == -3) return -1;
    return 0;
}

/* env_cython.pyx:164-165 ; TTT/env.py:40-41 ; NBTTT/utils.py:238-245 */
int orc_env_winner(const orc_env *e) {
    if (e->game == ORC_C4) return c4_winner(e->cell);
    if (e->game == ORC_TTT) return ttt_winner(e->cell);
    for (int i = 1; i < 10; i++) {
        int w = nbttt_single(&e->cell[i * 10 + 1]);
        if (w != 0) return w;
    }
    return 0;
}

/* env_cython.pyx:76-77 ; TTT/utils.py:13-15.
 * NBTTT: BUILD-DEFINED: "full" = the CURRENT sub-board has no
 * empty cell, following the repo's own C agent (NBTTT/dependency/Negamax.c:34, utils.c:88-97).  The
 * reference env.py:49-50 (all 81 cells) would leave a position with zero legal moves non-terminal. */
int orc_env_full(const orc_env *e) {
    if (e->game == ORC_C4) {
        for (int i = 0; i < 42; i++)
            if (e->cell[i] == 0) return 0;
        return 1;
    }
    if (e->game == ORC_TTT) {
        for (int i = 0; i < 9; i++)
            if (e->cell[i] == 0) return 0;
        return 1;
    }
    for (int c = 1; c < 10; c++)
        if (e->cell[(e->curr + 1) * 10 + c] == 2) return 0;
    return 1;
}

/* env_cython.pyx:117-118 ; TTT/env.py:21-22 ; NBTTT/env.py:29-30 */
int orc_env_done(const orc_env *e) { return orc_env_full(e) || orc_env_winner(e) != 0; }

/* env_cython.pyx:120-121 ; TTT/utils.py:32-34 ; NBTTT/utils.py:248-256 -- ascending */
int orc_env_valid(const orc_env *e, int *out) {
    int n = 0;
    if (e->game == ORC_C4) {
        for (int c = 0; c < 7; c++) {
            int has = 0;
            for (int r = 0; r < 6; r++)
                if (e->cell[r * 7 + c] == 0) has = 1;
            if (has) out[n++] = c;
        }
    } else if (e->game == ORC_TTT) {
        for (int i = 0; i < 9; i++)
            if (e->cell[i] == 0) out[n++] = i;
    } else {
        for (int c = 1; c < 10; c++)
            if (e->cell[(e->curr + 1) * 10 + c] == 2) out[n++] = c - 1;
    }
    return n;
}

int orc_state_size(int game) { return game == ORC_C4 ? 3 * 42 : (game == ORC_TTT ? 27 : 28 * 9); }

/* env_cython.pyx:143-149 ; TTT/utils.py:5-11 ; NBTTT/utils.py:258-268 (plane index fixed, q17 (ii)) */
void orc_env_state(const orc_env *e, float *out) {
    if (e->game == ORC_C4 || e->game == ORC_TTT) {
        int n = e->game == ORC_C4 ? 42 : 9;
        for (int i = 0; i < n; i++) {
            out[i] = e->cell[i] == 1 ? 1.f : 0.f;
            out[n + i] = e->cell[i] == -1 ? 1.f : 0.f;
            out[2 * n + i] = (float)e->turn; /* pyx: +1/-1 ; TTT: turn scalar broadcast */
        }
    } else {
        memset(out, 0, sizeof(float) * 28 * 9);
        for (int b = 0; b < 9; b++)
            for (int k = 0; k < 9; k++) {
                int v = e->cell[(b + 1) * 10 + k + 1];
                if (v == 0) out[b * 9 + k] = 1.f;
                if (v == 1) out[(9 + b) * 9 + k] = 1.f;
            }
        for (int k = 0; k < 9; k++) out[(18 + e->curr) * 9 + k] = 1.f;
        for (int k = 0; k < 9; k++) out[27 * 9 + k] = (float)e->turn;
    }
}

/* env_cython.pyx:168-172 ; TTT/env.py:62-65 (np.fliplr).  NBTTT: no flip exists (q17 (iv)). */
void orc_env_flip(orc_env *e) {
    if (e->game == ORC_C4) {
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 3; c++) {
                int8_t t = e->cell[r * 7 + c];
                e->cell[r * 7 + c] = e->cell[r * 7 + 6 - c];
                e->cell[r * 7 + 6 - c] = t;
            }
    } else if (e->game == ORC_TTT) {
        for (int r = 0; r < 3; r++) {
            int8_t t = e->cell[r * 3];
            e->cell[r * 3] = e->cell[r * 3 + 2];
            e->cell[r * 3 + 2] = t;
        }
    }
}

/* env_cython.pyx:174-190 (identity on a mirror-symmetric board) ; TTT/env.py:67-69 (always mirrors) */
int orc_env_flip_action(const orc_env *e, int a) {
    if (e->game == ORC_C4) {
        int symmetric = 1;
        for (int r = 0; r < 6 && symmetric; r++)
            for (int c = 0; c < 3; c++)
                if (e->cell[r * 7 + c] != e->cell[r * 7 + 6 - c]) {
                    symmetric = 0;
                    break;
                }
        return symmetric ? a : 6 - a;
    } else if (e->game == ORC_TTT) {
        return (a / 3) * 3 + (2 - a % 3);
    }
    return a;
}

/* NBTTT/env.py:15-23 with the three np.random draws passed in explicitly (q17 (iii)):
 * board = first sub-board, a1 = first cell, pick = index into the remaining legal list. */
int orc_nbttt_open(orc_env *e, int board, int a1, int pick) {
    orc_env_reset(e, ORC_NBTTT);
    e->curr = board;
    int init = e->curr;
    orc_env_step(e, a1);
    int valid[9], n = orc_env_valid(e, valid), m = 0, v2[9];
    for (int i = 0; i < n; i++)
        if (valid[i] != init) v2[m++] = valid[i];
    int a2 = v2[pick % m];
    orc_env_step(e, a2);
    return a2;
}

/* Packed bitboard words as defined in include/azamd.h (the product's public state format).
 * Used to compare device results with the oracle and to key the hashed evaluator. */
int orc_state_words(int game) { return game == ORC_C4 ? 2 : (game == ORC_TTT ? 1 : 4); }

void orc_env_pack(const orc_env *e, uint64_t *w) {
    uint64_t tf = (e->turn == 1) ? 0ull : (1ull << 63);
    if (e->game == ORC_C4) {
        w[0] = w[1] = 0;
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 7; c++) {
                int bit = c * 7 + (5 - r);
                if (e->cell[r * 7 + c] == 1) w[0] |= 1ull << bit;
                if (e->cell[r * 7 + c] == -1) w[1] |= 1ull << bit;
            }
        w[1] |= tf;
    } else if (e->game == ORC_TTT) {
        w[0] = 0;
        for (int i = 0; i < 9; i++) {
            if (e->cell[i] == 1) w[0] |= 1ull << i;
            if (e->cell[i] == -1) w[0] |= 1ull << (16 + i);
        }
        w[0] |= tf;
    } else {
        w[0] = w[1] = w[2] = 0;
        for (int b = 0; b < 9; b++)
            for (int k = 0; k < 9; k++) {
                int v = e->cell[(b + 1) * 10 + k + 1];
                int sh = 18 * (b % 3);
                if (v == 0) w[b / 3] |= 1ull << (sh + k);
                if (v == 1) w[b / 3] |= 1ull << (sh + 9 + k);
            }
        w[3] = (uint64_t)e->curr | tf;
    }
}

void orc_env_unpack(orc_env *e, int game, const uint64_t *w) {
    orc_env_reset(e, game);
    if (game == ORC_C4) {
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 7; c++) {
                int bit = c * 7 + (5 - r);
                if ((w[0] >> bit) & 1) e->cell[r * 7 + c] = 1;
                if ((w[1] >> bit) & 1) e->cell[r * 7 + c] = -1;
            }
        e->turn = (w[1] >> 63) ? -1 : 1;
    } else if (game == ORC_TTT) {
        for (int i = 0; i < 9; i++) {
            if ((w[0] >> i) & 1) e->cell[i] = 1;
            if ((w[0] >> (16 + i)) & 1) e->cell[i] = -1;
        }
        e->turn = (w[0] >> 63) ? -1 : 1;
    } else {
        for (int b = 0; b < 9; b++)
            for (int k = 0; k < 9; k++) {
                int sh = 18 * (b % 3);
                if ((w[b / 3] >> (sh + k)) & 1) e->cell[(b + 1) * 10 + k + 1] = 0;
                if ((w[b / 3] >> (sh + 9 + k)) & 1) e->cell[(b + 1) * 10 + k + 1] = 1;
            }
        e->curr = (int)(w[3] & 15);
        e->turn = (w[3] >> 63) ? -1 : 1;
    }
}

/* ------------------------------------------------------------------ evaluators */

/* Synthetic deterministic evaluator ("hashed"): a stand-in for CNN.predict (Network.py:78-85) that
 * both the reference (through a fake policy_value_fn in make_fixtures.py), this oracle and the HIP
 * kernels can compute bit-identically.  probs sum to ~1 over ALL actions (like a softmax), value in [-1,1). */
void orc_eval_hash(const orc_env *aug, uint64_t seed, float *probs, float *value) {
    uint64_t w[4];
    int nw = orc_state_words(aug->game), A = orc_n_actions(aug->game);
    orc_env_pack(aug, w);
    uint64_t h = seed;
    for (int i = 0; i < nw; i++) h = mix64(h ^ w[i]);
    float raw[9], s = 0.f;
    for (int a = 0; a < A; a++) {
        raw[a] = (float)((mix64(h + (uint64_t)a + 1) >> 40) + 1) * (1.0f / 16777216.0f);
        s = s + raw[a];
    }
    for (int a = 0; a < A; a++) probs[a] = raw[a] / s;
    *value = (float)(mix64(h + 64) >> 40) * (1.0f / 8388608.0f) - 1.0f;
}

static void orc_eval_uniform(const orc_env *aug, float *probs, float *value) {
    int A = orc_n_actions(aug->game);
    float p = 1.0f / (float)A;
    for (int a = 0; a < A; a++) probs[a] = p;
    *value = 0.f;
}

/* ------------------------------------------------------------------ tree search */

static int node_new(orc_mcts *m, int parent, int action, float prior, float noise) {
    if (m->n_nodes == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 1024;
        m->nodes = (orc_node *)realloc(m->nodes, sizeof(orc_node) * (size_t)m->cap);
    }
    orc_node *nd = &m->nodes[m->n_nodes];
    nd->parent = parent; /* MCTS.py:10-19 */
    nd->first = -1;
    nd->nchild = 0;
    nd->action = action;
    nd->N = 0;
    nd->Q = 0.f;
    nd->P = prior;
    nd->noise = noise;
    return m->n_nodes++;
}

void orc_mcts_reset_tree(orc_mcts *m) { /* MCTS.py:92,135: TreeNode(None, 1) */
    m->n_nodes = 0;
    m->root = node_new(m, -1, -1, 1.0f, 1.0f);
}

orc_mcts *orc_mcts_new(int game, double c_init, int n_playout, int eval_kind, uint64_t eval_seed) {
    orc_mcts *m = (orc_mcts *)calloc(1, sizeof(orc_mcts));
    m->game = game;
    m->c_init = c_init;
    m->c_base = 500; /* MCTS.py:95 */
    m->n_playout = n_playout;
    m->eval_kind = eval_kind;
    m->eval_seed = eval_seed;
    m->flip_mode = ORC_FLIP_NEVER;
    orc_mcts_reset_tree(m);
    return m;
}

void orc_mcts_free(orc_mcts *m) {
    if (!m) return;
    free(m->nodes);
    free(m);
}

void orc_mcts_set_net(orc_mcts *m, const void *net) { m->net = net; }

void orc_mcts_set_flips(orc_mcts *m, int mode, const uint8_t *bits, int64_t n) {
    m->flip_mode = mode;
    m->flip_bits = bits;
    m->flip_len = n;
    m->flip_pos = 0;
}

void orc_mcts_set_inject(orc_mcts *m, const float *probs, const float *values, int64_t n) {
    m->inj_probs = probs;
    m->inj_values = values;
    m->inj_len = n;
    m->inj_pos = 0;
}

/* MCTS.py:47-55 with the float32 rounding sequence numpy>=2 produces:
 *   prior' = f32(0.75*prior) + f32(0.25*noise)
 *   A      = c_init + ln((1+Np+c_base)/c_base)            (float64)
 *   u      = f32( f32( f32(A)*prior' ) * f32(sqrt(Np)) ) / f32(1+n)
 *   score  = -Q + u                                        (float32)
 * Compiled with -ffp-contract=off so none of these fuse. */
float orc_puct(double c_init, int c_base, int Np, int n, float Q, float prior, float noise) {
    float pr = 0.75f * prior;
    float nz = 0.25f * noise;
    float p = pr + nz;
    if (n == 0) return INFINITY; /* -Q + inf; Q is 0 for an unvisited node */
    double A = c_init + log((double)(1 + Np + c_base) / (double)c_base);
    float t = (float)A * p;
    t = t * (float)sqrt((double)Np);
    float u = t / (float)(1 + n);
    return -Q + u;
}

/* MCTS.py:79-80: max() keeps the FIRST maximal child in insertion order. */
static int select_child(const orc_mcts *m, int node) {
    const orc_node *nd = &m->nodes[node];
    int best = -1;
    float bs = 0.f;
    for (int i = 0; i < nd->nchild; i++) {
        const orc_node *c = &m->nodes[nd->first + i];
        float s = orc_puct(m->c_init, m->c_base, nd->N, c->N, c->Q, c->P, c->noise);
        if (best < 0 || s > bs) {
            best = nd->first + i;
            bs = s;
        }
    }
    return best;
}

/* MCTS.py:82-87: recursive update, parent first with the negated value; float32 running mean. */
static void node_update(orc_mcts *m, int node, float v) {
    /* The reference recurses parent-first; the order is irrelevant (disjoint nodes). */
    while (node >= 0) {
        orc_node *nd = &m->nodes[node];
        nd->N += 1;
        float d = v - nd->Q;
        d = d / (float)nd->N;
        nd->Q = nd->Q + d;
        v = -v;
        node = nd->parent;
    }
}

static int next_flip(orc_mcts *m) {
    if (m->game == ORC_NBTTT) return 0; /* no flip augmentation exists for NBTTT */
    if (m->flip_mode == ORC_FLIP_STREAM) {
        int b = 0;
        if (m->flip_pos < m->flip_len) b = m->flip_bits[m->flip_pos];
        m->flip_pos++;
        return b;
    }
    return 0;
}

/* MCTS.py:139-164 -- one playout. */
void orc_mcts_playout(orc_mcts *m, const orc_env *root_env) {
    orc_env env = *root_env; /* MCTS.py:169 env.copy() */
    int node = m->root;
    int depth = 0;
    while (m->nodes[node].nchild > 0) { /* MCTS.py:108-113 */
        node = select_child(m, node);
        orc_env_step(&env, m->nodes[node].action);
        depth++;
    }
    m->stat_levels += depth;
    m->stat_playouts += 1;
    int flipped = next_flip(m); /* MCTS.py:142: the random draw happens on every playout */
    orc_env aug = env;
    if (flipped) orc_env_flip(&aug);
    int valid[9];
    int nv = orc_env_valid(&aug, valid); /* MCTS.py:144 */
    int done = orc_env_done(&env);
    float probs[9], value = 0.f;
    if (m->eval_kind == ORC_EVAL_INJECT) {
        int A = orc_n_actions(m->game);
        if (m->inj_pos < m->inj_len) {
            memcpy(probs, m->inj_probs + m->inj_pos * A, sizeof(float) * (size_t)A);
            value = m->inj_values[m->inj_pos];
        } else {
            orc_eval_uniform(&aug, probs, &value);
        }
        m->inj_pos++;
    } else if (m->eval_kind == ORC_EVAL_UNIFORM) {
        orc_eval_uniform(&aug, probs, &value);
    } else if (m->eval_kind == ORC_EVAL_HASH) {
        orc_eval_hash(&aug, m->eval_seed, probs, &value);
    } else { /* ORC_EVAL_CNN.  q7: the reference also evaluates terminal leaves and discards the result. */
        if (!done) {
            float st[3 * 42];
            orc_env_state(&aug, st);
            orc_cnn_predict(m->net, st, probs, &value);
            m->stat_evals += 1;
        }
    }
    /* MCTS.py:147-148: probs[valid]; probs /= sum(probs)  (left-to-right float32 sum) */
    float pv[9], s = 0.f;
    for (int i = 0; i < nv; i++) {
        pv[i] = probs[valid[i]];
        s = s + pv[i];
    }
    for (int i = 0; i < nv; i++) pv[i] = pv[i] / s;
    int acts[9];
    for (int i = 0; i < nv; i++) acts[i] = flipped ? orc_env_flip_action(&env, valid[i]) : valid[i]; /* :152-153 */
    float leaf_value = value;
    if (!done) { /* MCTS.py:154-157 ; TreeNode.expand :71-77 */
        orc_node *nd;
        int first = m->n_nodes;
        for (int i = 0; i < nv; i++) node_new(m, node, acts[i], pv[i], pv[i]);
        nd = &m->nodes[node];
        nd->first = first;
        nd->nchild = nv;
    } else { /* MCTS.py:158-163 */
        int w = orc_env_winner(&env);
        leaf_value = (w == 0) ? 0.f : ((w == env.turn) ? 1.f : -1.f);
    }
    node_update(m, node, leaf_value); /* MCTS.py:164 */
}

/* MCTS.py:166-171: S playouts, then (action, visits) of the root children in insertion order. */
int orc_mcts_get_action_visits(orc_mcts *m, const orc_env *env, int *actions, int *visits) {
    for (int i = 0; i < m->n_playout; i++) orc_mcts_playout(m, env);
    const orc_node *r = &m->nodes[m->root];
    for (int i = 0; i < r->nchild; i++) {
        actions[i] = m->nodes[r->first + i].action;
        visits[i] = m->nodes[r->first + i].N;
    }
    return r->nchild;
}

/* MCTS.py:130-135.  The oracle never frees the discarded part of the tree (test sizes are small). */
void orc_mcts_prune_root(orc_mcts *m, int action) {
    const orc_node *r = &m->nodes[m->root];
    for (int i = 0; i < r->nchild; i++)
        if (m->nodes[r->first + i].action == action) {
            m->root = r->first + i;
            m->nodes[m->root].parent = -1;
            return;
        }
    orc_mcts_reset_tree(m);
}

int orc_mcts_root_visits(const orc_mcts *m) { return m->nodes[m->root].N; }
float orc_mcts_root_q(const orc_mcts *m) { return m->nodes[m->root].Q; }
int orc_mcts_n_nodes(const orc_mcts *m) { return m->n_nodes; }
double orc_mcts_mean_levels(const orc_mcts *m) { return m->stat_playouts ? (double)m->stat_levels / (double)m->stat_playouts : 0.0; }
int64_t orc_mcts_stat_playouts(const orc_mcts *m) { return m->stat_playouts; }
int64_t orc_mcts_stat_evals(const orc_mcts *m) { return m->stat_evals; }

/* child stats of an arbitrary path, for node-level comparisons with the device tree:
 * path = list of actions from the root; returns number of children and fills arrays. */
int orc_mcts_children(const orc_mcts *m, const int *path, int plen, int *actions, int *visits, float *q, float *p) {
    int node = m->root;
    for (int d = 0; d < plen; d++) {
        const orc_node *nd = &m->nodes[node];
        int nxt = -1;
        for (int i = 0; i < nd->nchild; i++)
            if (m->nodes[nd->first + i].action == path[d]) nxt = nd->first + i;
        if (nxt < 0) return -1;
        node = nxt;
    }
    const orc_node *nd = &m->nodes[node];
    for (int i = 0; i < nd->nchild; i++) {
        const orc_node *c = &m->nodes[nd->first + i];
        actions[i] = c->action;
        visits[i] = c->N;
        q[i] = c->Q;
        p[i] = c->P;
    }
    return nd->nchild;
}

/* ------------------------------------------------------------------ move selection */

/* numpy's float64 add-reduce order for a short contiguous array: plain loop below 8 elements,
 * otherwise 8 interleaved partial sums combined pairwise, then the tail. */
static double np_sum(const double *a, int n) {
    if (n < 8) {
        double r = 0.0; /* numpy starts from -0.0/identity; adding to 0.0 is value-identical here */
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    for (int i = 0; i < 8; i++) r[i] = a[i];
    int i = 8;
    for (; i + 8 <= n; i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* player.py:101-111 + utils.py:55-57.  pi is float32[n_actions] (zeros where no child).
 * Returns the index (into actions[]) of the deterministic parity-mode choice: the first maximal
 * entry of the sampling distribution (np.argmax semantics), which is what make_fixtures.py
 * substitutes for np.random.choice. */
int orc_move_probs(const int *actions, const int *visits, int n, double temp, int n_actions, float *pi) {
    double x[9], ex[9], t = temp > 1e-8 ? temp : 1e-8, mx = -INFINITY;
    for (int i = 0; i < n_actions; i++) pi[i] = 0.f;
    for (int i = 0; i < n; i++) {
        x[i] = log((double)visits[i]) / t;
        if (x[i] > mx) mx = x[i];
    }
    for (int i = 0; i < n; i++) ex[i] = exp(x[i] - mx);
    double s = np_sum(ex, n);
    int best = 0;
    for (int i = 0; i < n; i++) {
        double d = ex[i] / s;
        pi[actions[i]] = (float)d;
    }
    if (temp == 0) { /* player.py:106-108: uniform over the maxima */
        int mv = visits[0];
        for (int i = 1; i < n; i++)
            if (visits[i] > mv) mv = visits[i];
        for (int i = 0; i < n; i++)
            if (visits[i] == mv) {
                best = i;
                break;
            }
    } else {
        double bd = -1.0;
        for (int i = 0; i < n; i++) {
            double d = ex[i] / s;
            if (d > bd) {
                bd = d;
                best = i;
            }
        }
    }
    return best;
}

/* ------------------------------------------------------------------ game loops */

/* game.py:35-58 (self-play with tree reuse when reuse_tree!=0: player.py:112-115) and
 * game.py:12-33 (play.py --sp: temp 0, fresh tree each ply) with the deterministic move rule.
 * Outputs per ply: packed pre-move state words, pi, mover, action; returns number of plies.
 * z (mover-relative outcome, game.py:52-55) is filled for every ply at the end. */
int orc_selfplay_game(orc_mcts *m, const orc_env *start, double temp, int first_n_steps, int reuse_tree, int max_plies,
                      uint64_t *states, float *pis, int *players, int *actions_out, float *z, int *winner_out,
                      int *visits_out /* [plies][n_actions] or NULL */) {
    orc_env env = *start;
    int A = orc_n_actions(m->game), nw = orc_state_words(m->game), steps = 0;
    orc_mcts_reset_tree(m);
    while (steps < max_plies) {
        double temperature = (first_n_steps >= 0) ? ((steps >= first_n_steps) ? 1e-3 : temp) : temp; /* game.py:41 */
        int acts[9], vis[9];
        int n = orc_mcts_get_action_visits(m, &env, acts, vis);
        int pick = orc_move_probs(acts, vis, n, temperature, A, pis + (size_t)steps * A);
        int action = acts[pick];
        if (visits_out) {
            for (int a = 0; a < A; a++) visits_out[(size_t)steps * A + a] = 0;
            for (int i = 0; i < n; i++) visits_out[(size_t)steps * A + acts[i]] = vis[i];
        }
        if (reuse_tree)
            orc_mcts_prune_root(m, action);
        else
            orc_mcts_reset_tree(m);
        orc_env_pack(&env, states + (size_t)steps * nw);
        players[steps] = env.turn;
        actions_out[steps] = action;
        steps++;
        orc_env_step(&env, action);
        if (orc_env_done(&env)) break;
    }
    int w = orc_env_winner(&env);
    *winner_out = w;
    for (int i = 0; i < steps; i++) z[i] = (w == 0) ? 0.f : ((players[i] == w) ? 1.f : -1.f);
    return steps;
}
