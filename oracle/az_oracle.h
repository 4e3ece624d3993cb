/* az_oracle.h -- declarations of the CPU oracle (TEST INFRASTRUCTURE ONLY, see az_oracle.c). */
#ifndef AZ_ORACLE_H
#define AZ_ORACLE_H
#include <stdint.h>

#define ORC_C4 0
#define ORC_TTT 1
#define ORC_NBTTT 2

#define ORC_EVAL_UNIFORM 0
#define ORC_EVAL_HASH 1
#define ORC_EVAL_CNN 2
#define ORC_EVAL_INJECT 3

#define ORC_FLIP_NEVER 0
#define ORC_FLIP_STREAM 1

/* The reference's own representation: cell arrays.
 * C4: cell[r*7+c], r=0 top row, values +1 (X) / -1 (O) / 0     (env_cython.pyx:18)
 * TTT: cell[r*3+c], same values                                 (TTT/env.py:8)
 * NBTTT: cell[b*10+k] = boards[b][k], 1-based, 2 empty/0 X/1 O  (NBTTT/env.py:13,39-47) */
typedef struct {
    int32_t game, turn, curr, pad;
    int8_t cell[104];
} orc_env;

typedef struct {
    int32_t parent, first, nchild, action, N;
    float Q, P, noise;
} orc_node;

typedef struct {
    orc_node *nodes;
    int32_t n_nodes, cap, root;
    int32_t game;
    double c_init;
    int32_t c_base, n_playout;
    int32_t eval_kind;
    uint64_t eval_seed;
    const void *net;
    int32_t flip_mode;
    const uint8_t *flip_bits;
    int64_t flip_len, flip_pos;
    const float *inj_probs, *inj_values;
    int64_t inj_len, inj_pos;
    int64_t stat_levels, stat_playouts, stat_evals;
} orc_mcts;

int orc_n_actions(int game);
void orc_env_reset(orc_env *e, int game);
void orc_env_from_cells(orc_env *e, int game, const int8_t *cells, int n);
int orc_env_place(orc_env *e, int action);
void orc_env_step(orc_env *e, int action);
int orc_env_winner(const orc_env *e);
int orc_env_full(const orc_env *e);
int orc_env_done(const orc_env *e);
int orc_env_valid(const orc_env *e, int *out);
int orc_state_size(int game);
void orc_env_state(const orc_env *e, float *out);
void orc_env_flip(orc_env *e);
int orc_env_flip_action(const orc_env *e, int a);
int orc_nbttt_open(orc_env *e, int board, int a1, int pick);
int orc_state_words(int game);
void orc_env_pack(const orc_env *e, uint64_t *w);
void orc_env_unpack(orc_env *e, int game, const uint64_t *w);

void orc_eval_hash(const orc_env *aug, uint64_t seed, float *probs, float *value);

orc_mcts *orc_mcts_new(int game, double c_init, int n_playout, int eval_kind, uint64_t eval_seed);
void orc_mcts_free(orc_mcts *m);
void orc_mcts_reset_tree(orc_mcts *m);
void orc_mcts_set_net(orc_mcts *m, const void *net);
void orc_mcts_set_flips(orc_mcts *m, int mode, const uint8_t *bits, int64_t n);
void orc_mcts_set_inject(orc_mcts *m, const float *probs, const float *values, int64_t n);
float orc_puct(double c_init, int c_base, int Np, int n, float Q, float prior, float noise);
void orc_mcts_playout(orc_mcts *m, const orc_env *root_env);
int orc_mcts_get_action_visits(orc_mcts *m, const orc_env *env, int *actions, int *visits);
void orc_mcts_prune_root(orc_mcts *m, int action);
int orc_mcts_root_visits(const orc_mcts *m);
float orc_mcts_root_q(const orc_mcts *m);
int orc_mcts_n_nodes(const orc_mcts *m);
double orc_mcts_mean_levels(const orc_mcts *m);
int64_t orc_mcts_stat_playouts(const orc_mcts *m);
int64_t orc_mcts_stat_evals(const orc_mcts *m);
int orc_mcts_children(const orc_mcts *m, const int *path, int plen, int *actions, int *visits, float *q, float *p);
int orc_move_probs(const int *actions, const int *visits, int n, double temp, int n_actions, float *pi);
int orc_selfplay_game(orc_mcts *m, const orc_env *start, double temp, int first_n_steps, int reuse_tree, int max_plies,
                      uint64_t *states, float *pis, int *players, int *actions_out, float *z, int *winner_out,
                      int *visits_out);

/* az_oracle_pure.c: the pure-MCTS (UCT + rollout) evaluation opponent, MCTS.py:57-63,115-128, utils.py:26-52 */
typedef struct orc_pure orc_pure;
orc_pure *orc_pure_new(int game, double c_init, int n_playout, uint64_t seed);
void orc_pure_free(orc_pure *m);
void orc_pure_reset_tree(orc_pure *m);
void orc_pure_set_flips(orc_pure *m, int mode, const uint8_t *bits, int64_t n);
double orc_uct(double c_init, int c_base, int Np, int n, double Q);
int orc_rollout_pick(const orc_env *e, uint64_t seed, int nv);
int orc_rollout(const orc_env *env, uint64_t seed, int64_t *plies_out);
void orc_pure_playout(orc_pure *m, const orc_env *root_env);
int orc_pure_get_action(orc_pure *m, const orc_env *env, int *actions, int *visits, double *q, int *n_out);
void orc_pure_prune_root(orc_pure *m, int action);
int orc_pure_root_visits(const orc_pure *m);
double orc_pure_root_q(const orc_pure *m);
int64_t orc_pure_stat_playouts(const orc_pure *m);
int64_t orc_pure_stat_levels(const orc_pure *m);
int64_t orc_pure_stat_rollout_plies(const orc_pure *m);

/* az_oracle_cnn.c */
#define ORC_CNN_BLOB_FLOATS 500106
void *orc_cnn_new(const float *blob, int64_t n_floats);
void orc_cnn_free(void *net);
void orc_cnn_forward(const void *net, const float *state /* [3][6][7] */, float *logp /*7*/, float *logv /*3*/);
void orc_cnn_predict(const void *net, const float *state, float *probs /*7*/, float *value /*1*/);

#endif
