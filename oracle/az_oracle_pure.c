/* az_oracle_pure.c -- CPU restatement of the reference's PURE Monte-Carlo tree search (no network): the evaluation
 * opponent of the Elo arena (Pipeline.py:125-135 -> player.MCTSPlayer, player.py:71-82).
 *
 * TEST INFRASTRUCTURE ONLY (see az_oracle.c).  Follows, in the reference's own representation:
 *   TreeNode.UCT            MCTS.py:57-63     all Python floats -> float64 arithmetic, math.log / math.sqrt
 *   TreeNode.select/update  MCTS.py:79-87     first maximum in insertion order; Q += (v - Q) / n in float64
 *   MCTS.playout            MCTS.py:115-123   UCT walk, random_flip, policy on the (mirrored) leaf, expand unless done
 *   MCTS.get_action         MCTS.py:125-128   n_playout playouts, then the most visited root child (first maximum)
 *   policy_value_fn         utils.py:43-46    uniform priors over the valid moves (unused by UCT) + a rollout
 *   evaluate_rollout        utils.py:26-40    play to the end, value from the leaf mover's point of view
 *   rollout_policy_fn       utils.py:49-52    np.random.rand per valid move, arg-max = a uniformly random valid move
 * The rollout's random source is the one thing the harness owns (make_fixtures_pure.py patches rollout_policy_fn):
 * the move index is a hash of the position, idx = (mix64(fold(words, seed) + 'ROLL') >> 33) % n_valid, so that the
 * reference, this oracle and the HIP kernel play identical rollouts. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "az_oracle.h"

typedef struct {
    int32_t parent, first, nchild, action, N;
    double Q;
} pure_node;

struct orc_pure {
    pure_node *nodes;
    int32_t n_nodes, cap, root, game;
    double c_init;
    int32_t c_base, n_playout;
    uint64_t seed;
    int32_t flip_mode;
    const uint8_t *flip_bits;
    int64_t flip_len, flip_pos;
    int64_t stat_levels, stat_playouts, stat_rollout_plies;
};

static uint64_t mix64(uint64_t z) { /* splitmix64 finaliser (public domain constants) */
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int pnode_new(orc_pure *m, int parent, int action) {
    if (m->n_nodes == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 1024;
        m->nodes = (pure_node *)realloc(m->nodes, sizeof(pure_node) * (size_t)m->cap);
    }
    pure_node *nd = &m->nodes[m->n_nodes];
    nd->parent = parent;
    nd->first = -1;
    nd->nchild = 0;
    nd->action = action;
    nd->N = 0;
    nd->Q = 0.0;
    return m->n_nodes++;
}

void orc_pure_reset_tree(orc_pure *m) {
    m->n_nodes = 0;
    m->root = pnode_new(m, -1, -1);
}

orc_pure *orc_pure_new(int game, double c_init, int n_playout, uint64_t seed) {
    orc_pure *m = (orc_pure *)calloc(1, sizeof(orc_pure));
    m->game = game;
    m->c_init = c_init;
    m->c_base = 500; /* MCTS.py:95 */
    m->n_playout = n_playout;
    m->seed = seed;
    orc_pure_reset_tree(m);
    return m;
}

void orc_pure_free(orc_pure *m) {
    if (!m) return;
    free(m->nodes);
    free(m);
}

void orc_pure_set_flips(orc_pure *m, int mode, const uint8_t *bits, int64_t n) {
    m->flip_mode = mode;
    m->flip_bits = bits;
    m->flip_len = n;
    m->flip_pos = 0;
}

/* MCTS.py:57-63, float64 throughout. */
double orc_uct(double c_init, int c_base, int Np, int n, double Q) {
    if (n == 0) return INFINITY;
    double A = c_init + log((double)(1 + Np + c_base) / (double)c_base);
    double t = log((double)Np) / (double)n;
    double u = A * sqrt(t);
    return -Q + u;
}

/* the harness's rollout policy: index of the move to play among the ascending valid moves */
int orc_rollout_pick(const orc_env *e, uint64_t seed, int nv) {
    uint64_t w[4], h = seed;
    int nw = orc_state_words(e->game);
    orc_env_pack(e, w);
    for (int i = 0; i < nw; i++) h = mix64(h ^ w[i]);
    return (int)((mix64(h + 0x524F4C4Cull) >> 33) % (uint64_t)nv);
}

/* utils.py:26-40 on a copy of env. */
int orc_rollout(const orc_env *env, uint64_t seed, int64_t *plies_out) {
    orc_env e = *env;
    int player = e.turn, valid[9];
    int64_t plies = 0;
    for (int it = 0; it < 1000; it++) {
        if (orc_env_done(&e)) break;
        int nv = orc_env_valid(&e, valid);
        orc_env_step(&e, valid[orc_rollout_pick(&e, seed, nv)]);
        plies++;
    }
    if (plies_out) *plies_out = plies;
    int w = orc_env_winner(&e);
    return w == 0 ? 0 : (w == player ? 1 : -1);
}

static int pure_select(const orc_pure *m, int node) { /* MCTS.py:79-80 with UCT=True */
    const pure_node *nd = &m->nodes[node];
    int best = -1;
    double bs = 0.0;
    for (int i = 0; i < nd->nchild; i++) {
        const pure_node *c = &m->nodes[nd->first + i];
        double s = orc_uct(m->c_init, m->c_base, nd->N, c->N, c->Q);
        if (best < 0 || s > bs) {
            best = nd->first + i;
            bs = s;
        }
    }
    return best;
}

void orc_pure_playout(orc_pure *m, const orc_env *root_env) { /* MCTS.py:115-123 */
    orc_env env = *root_env;
    int node = m->root, depth = 0;
    while (m->nodes[node].nchild > 0) {
        node = pure_select(m, node);
        orc_env_step(&env, m->nodes[node].action);
        depth++;
    }
    m->stat_levels += depth;
    m->stat_playouts += 1;
    int flipped = 0;
    if (m->game != ORC_NBTTT && m->flip_mode == ORC_FLIP_STREAM) {
        if (m->flip_pos < m->flip_len) flipped = m->flip_bits[m->flip_pos];
        m->flip_pos++;
    }
    orc_env aug = env;
    if (flipped) orc_env_flip(&aug);
    int valid[9];
    int nv = orc_env_valid(&aug, valid); /* utils.py:44 */
    int64_t plies = 0;
    double leaf_value = (double)orc_rollout(&aug, m->seed, &plies); /* utils.py:46 */
    m->stat_rollout_plies += plies;
    if (!orc_env_done(&env)) { /* MCTS.py:119-122 */
        int first = m->n_nodes;
        for (int i = 0; i < nv; i++) pnode_new(m, node, flipped ? orc_env_flip_action(&env, valid[i]) : valid[i]);
        m->nodes[node].first = first;
        m->nodes[node].nchild = nv;
    }
    for (int n = node; n >= 0; n = m->nodes[n].parent) { /* MCTS.py:82-87 */
        pure_node *nd = &m->nodes[n];
        nd->N += 1;
        nd->Q += (leaf_value - nd->Q) / (double)nd->N;
        leaf_value = -leaf_value;
    }
}

/* MCTS.py:125-128: returns the chosen action (most visited, first maximum); fills the root children. */
int orc_pure_get_action(orc_pure *m, const orc_env *env, int *actions, int *visits, double *q, int *n_out) {
    for (int i = 0; i < m->n_playout; i++) orc_pure_playout(m, env);
    const pure_node *r = &m->nodes[m->root];
    int best = -1, bv = -1;
    for (int i = 0; i < r->nchild; i++) {
        const pure_node *c = &m->nodes[r->first + i];
        if (actions) actions[i] = c->action;
        if (visits) visits[i] = c->N;
        if (q) q[i] = c->Q;
        if (c->N > bv) {
            bv = c->N;
            best = c->action;
        }
    }
    if (n_out) *n_out = r->nchild;
    return best;
}

void orc_pure_prune_root(orc_pure *m, int action) { /* MCTS.py:130-135 */
    const pure_node *r = &m->nodes[m->root];
    for (int i = 0; i < r->nchild; i++)
        if (m->nodes[r->first + i].action == action) {
            m->root = r->first + i;
            m->nodes[m->root].parent = -1;
            return;
        }
    orc_pure_reset_tree(m);
}

int orc_pure_root_visits(const orc_pure *m) { return m->nodes[m->root].N; }
double orc_pure_root_q(const orc_pure *m) { return m->nodes[m->root].Q; }
int64_t orc_pure_stat_playouts(const orc_pure *m) { return m->stat_playouts; }
int64_t orc_pure_stat_levels(const orc_pure *m) { return m->stat_levels; }
int64_t orc_pure_stat_rollout_plies(const orc_pure *m) { return m->stat_rollout_plies; }
