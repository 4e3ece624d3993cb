/*
 * azamd.h -- C ABI of libazamd.so: the MI355X (gfx950) self-play engine.
 *
 * This is the drop-in boundary for the reference's self-play hot path
 * (env step + PUCT search + CNN leaf evaluation).  The reference has a single precedent for a
 * native boundary -- ctypes over plain int buffers (environments/NBTTT/utils.py:16,129-133 binding
 * dependency/Negamax.c:71-93) -- and this header follows the same style: plain pointers and sizes,
 * caller-owned host buffers, int status returns.  No torch types appear here.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (never abort/printf); az_last_error() gives text.
 *   - all pointers are HOST pointers unless the name ends in _dev; the engine owns device memory.
 *   - an engine is bound to one HIP device and is NOT thread-safe (one host thread per GPU, like the
 *     reference's single-threaded model).  Calls are synchronous on return unless stated otherwise.
 *   - without a usable HIP device every compute entry point fails (AZ_ERR_NO_DEVICE): there is no CPU path.
 *
 * Packed state words (uint64 little endian), `az_state_words(game)` words per position:
 *   Connect4 (2): w0 = X stones, w1 = O stones | bit63 (set <=> O to move, reference turn == -1).
 *                 stone bit = col*7 + h, h = 0 bottom row (reference row 5) .. 5 top row (row 0); bit col*7+6 = 0.
 *   TTT      (1): X bits 0..8 (bit = row*3+col), O bits 16..24, bit63 = O to move.
 *   NBTTT    (4): sub-board b in word b/3 at shift 18*(b%3): 9 X bits, then 9 O bits (bit = cell 0..8);
 *                 w3 = current sub-board (bits 0..3) | bit63 = O to move.
 * Reference equivalents: env_cython.pyx:21-22 (float board + turn), TTT/env.py:8-9, NBTTT/env.py:13-16.
 */
#ifndef AZAMD_H
#define AZAMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_GAME_C4 0
#define AZ_GAME_TTT 1
#define AZ_GAME_NBTTT 2

#define AZ_EVAL_UNIFORM 0 /* priors 1/A over all actions, value 0 (BASELINE config 2)                    */
#define AZ_EVAL_HASH 1    /* synthetic deterministic evaluator keyed by the evaluated position (parity)   */
#define AZ_EVAL_CNN 2     /* fused MFMA CNN forward (Connect4 only), weights from az_net_load            */
#define AZ_EVAL_HOST 3    /* caller supplies priors/values per leaf batch: az_select / az_backup          */
/* Pure Monte-Carlo tree search, the reference's MCTS class with utils.policy_value_fn (MCTS.py:57-63,90-128,
 * utils.py:26-52; the Elo arena's opponent player.MCTSPlayer): UCT selection in float64, value = one rollout. */
#define AZ_EVAL_ROLLOUT 4      /* rollout moves uniformly random over the valid moves (device RNG)            */
#define AZ_EVAL_ROLLOUT_HASH 5 /* rollout move = hash of the position (parity with the fixture harness)       */

#define AZ_FLIP_NEVER 0  /* parity: np.random.rand patched to never flip                                  */
#define AZ_FLIP_STREAM 1 /* parity: explicit bit per playout, az_set_flip_stream                          */
#define AZ_FLIP_RANDOM 2 /* production: counter-based device RNG, p = 0.5 (env_cython.pyx:192-199)        */

#define AZ_MOVE_FIRST_MAX 0 /* parity: first maximum of the sampling distribution, insertion order       */
#define AZ_MOVE_SAMPLE 1    /* production: sample (player.py:106-111)                                     */

#define AZ_MAX_COHORTS 8

#define AZ_OK 0
#define AZ_ERR_ARG (-1)
#define AZ_ERR_NO_DEVICE (-2)
#define AZ_ERR_HIP (-3)
#define AZ_ERR_STATE (-4)
#define AZ_ERR_ARENA (-5) /* a per-game node arena overflowed; search results are incomplete             */
#define AZ_ERR_NO_NET (-6)

typedef struct az_engine az_engine;

/* Engine configuration.  Mirrors the constructor arguments of MCTS_AZ / AlphaZeroPlayer
 * (MCTS.py:91-98, player.py:86-90) plus the batch shape. */
typedef struct az_config {
    int32_t game;           /* AZ_GAME_*                                                                  */
    int32_t device;         /* HIP device ordinal                                                         */
    int32_t n_games;        /* G: concurrent game slots (one tree each)                                   */
    int32_t n_playout;      /* S: playouts per move (MCTS.py:96)                                          */
    double c_init;          /* c_puct (MCTS.py:94), float64 like the reference's Python float             */
    int32_t c_base;         /* 500 (MCTS.py:95)                                                           */
    int32_t evaluator;      /* AZ_EVAL_*                                                                  */
    uint64_t eval_seed;     /* AZ_EVAL_HASH key                                                           */
    int32_t flip_mode;      /* AZ_FLIP_*                                                                  */
    int32_t reuse_tree;     /* 1: prune_root(action) keeps the subtree (player.py:112-113); 0: fresh tree */
    double dirichlet_alpha; /* <=0: eval() mode, no noise.  >0: train() mode, noise at EVERY expansion    */
    uint64_t rng_seed;      /* device RNG seed (flips, Dirichlet, move sampling)                          */
    int32_t move_mode;      /* AZ_MOVE_*                                                                  */
    int32_t max_plies;      /* per-game ply cap for arena sizing; 0 = game maximum                        */
    int64_t nodes_per_game; /* arena capacity in nodes; 0 = 1 + A*S*(reuse ? max_plies : 1)               */
    uint64_t game_id_base;  /* global id of the first game (slot 0); keys the RNG streams; see game_id_stride */
    int32_t n_cohorts;      /* AZ_EVAL_CNN: slot ranges searched on their own HIP streams so that one range's  */
                            /* tree kernels overlap the other's CNN launch; 0 = auto (4 from 2048 slots [4096 without the evaluation cache], 3 from 768, 2 from 512), max 8. */
                            /* Results do not depend on it (games are independent).                            */
    uint64_t game_id_stride; /* the k-th game this engine starts (slots first, then refills) has the global id      */
                            /* game_id_base + game_id_stride * k; 0 = 1.  Multi-GPU: base = rank, stride = world size, */
                            /* so the ids 0..M-1 of a job's first M games are the same however the job is split.      */
    int64_t eval_cache_entries; /* AZ_EVAL_CNN: entries (64 B each, rounded up to a power of two) of the evaluation cache, a     */
                            /* device table shared by all the games of the engine that answers a leaf whose position the     */
                            /* network has already evaluated with the stored probs and value -- the very bits K3 delivers   */
                            /* (CNN.predict is a pure function of the position, Network.py:78-85), so every game's           */
                            /* trajectory is the same with the cache on or off; the reference evaluates every leaf anew      */
                            /* (MCTS.py:146).  0 = auto (128 Ki entries = 8 MB per slot, 2^16 .. 2^30, and at most half of the */
                            /* device memory free at creation), < 0 = off; a size given here is allocated as given or the     */
                            /* engine is refused.  az_net_load empties it.  az_get_config reports the size in use (-1: none). */
} az_config;

typedef struct az_stats {
    int64_t playouts;      /* playouts executed                                                           */
    int64_t levels;        /* tree levels walked by select (sum of depths)                                */
    int64_t leaf_evals;    /* non-terminal leaves the evaluator (K3 / host / synthetic) evaluated          */
    int64_t terminal_hits; /* playouts that ended on a terminal node                                      */
    int64_t children;      /* child records written by expansion                                          */
    int64_t nodes_used;    /* sum over games of arena nodes in use                                        */
    int64_t games_finished;
    int64_t plies;
    int64_t rollout_plies; /* moves played inside rollouts (AZ_EVAL_ROLLOUT*)                             */
    int64_t cache_hits;    /* non-terminal leaves answered by the evaluation cache instead of the network;  */
                           /* leaf_evals + cache_hits + terminal_hits == playouts                           */
} az_stats;

const char *az_last_error(void);
int az_version(void);
int az_state_words(int game);
int az_n_actions(int game);
int az_device_count(void); /* 0 when no HIP device is usable */

/* ---- K1: batched environment kernels (stateless).  Replace env_cython.pyx:76-141 / TTT/utils.py:13-45 /
 *      NBTTT/env.py:39-60 for n independent positions.  Any output pointer may be NULL. ---------------- */
int az_env_step_batch(int game, int device, int64_t n, const uint64_t *states_in, const int32_t *actions,
                      uint64_t *states_out, uint8_t *placed, int8_t *winner, uint8_t *done, uint16_t *legal);
/* status only (no move): winner (winPlayer), done, legal mask (valid_mask), full (check_full). */
int az_env_status_batch(int game, int device, int64_t n, const uint64_t *states, int8_t *winner, uint8_t *done,
                        uint16_t *legal, uint8_t *full);
/* network input planes (env.current_state): Connect4 (n,3,6,7), TTT (n,3,3,3), NBTTT (n,28,3,3) float32. */
int az_env_planes_batch(int game, int device, int64_t n, const uint64_t *states, float *planes);
/* mirror augmentation: flipped states (env.flip) and the action map env.flip_action for every action. */
int az_env_flip_batch(int game, int device, int64_t n, const uint64_t *states, uint64_t *flipped,
                      int8_t *flip_action /* [n][A] */);
/* device-resident variant used by bench.py / rocprof (pointers are DEVICE memory, launch is async on stream 0). */
int az_env_step_batch_dev(int game, int64_t n, const uint64_t *states_in_dev, const int32_t *actions_dev,
                          uint64_t *states_out_dev, uint8_t *placed_dev, int8_t *winner_dev, uint8_t *done_dev,
                          uint16_t *legal_dev);

/* ---- engine life cycle ------------------------------------------------------------------------------ */
int az_engine_create(const az_config *cfg, az_engine **out);
void az_engine_destroy(az_engine *e);
/* hipStreamSynchronize on the engine stream.  Also where the kernels' overflow flags surface: AZ_ERR_ARENA when an
 * expansion did not fit the node arena (that leaf stayed a leaf) or a finished game did not fit the trajectory
 * buffer (that game was dropped).  The flag is reported once and cleared; nothing is written out of bounds. */
int az_engine_sync(az_engine *e);
/* the configuration in force, with every 0 = auto field resolved (c_base, nodes_per_game, n_cohorts). */
int az_get_config(az_engine *e, az_config *out);
int az_get_stats(az_engine *e, az_stats *out);
int az_reset_stats(az_engine *e);
/* Live kernel timing with HIP events recorded on the engine's own stream around every search kernel launch.
 * Categories: 0 = CNN leaf evaluation (K3), 1 = select (K2), 2 = expand/backup (K2), 3 = fused search (K2).
 * az_get_profile returns accumulated milliseconds and launch counts since az_set_profiling(e, on);
 * on = 1: every category, 2: the CNN launches only (half the events to record and read back), 3: the CNN launches and, of
 * the other categories, every fourth launch of the first cohort only (a sample: 1 in 16 with four cohorts), 4: blocks of
 * 64 consecutive rounds out of every 512 -- inside a block every cohort's CNN launch and the first cohort's tree step,
 * outside it nothing (an event is a queue packet between two dependent kernels: ~6 us per gap), 0: off. */
int az_set_profiling(az_engine *e, int on);
int az_get_profile(az_engine *e, double *ms4, int64_t *launches4);
/* per category, the time (ms) during which at least one launch of it was executing: the union of the launches'
 * [start, stop] intervals.  Equals the sum of durations when launches never overlap (n_cohorts == 1). */
int az_get_profile_busy(az_engine *e, double *busy_ms4);
/* MCTS.train()/eval() (MCTS.py:100-106) + alpha: switch Dirichlet noise for FUTURE expansions. */
int az_set_noise(az_engine *e, double dirichlet_alpha);
int az_set_n_playout(az_engine *e, int32_t n_playout);

/* ---- roots and trees ---------------------------------------------------------------------------------- */
/* env.reset() for every slot + fresh trees (MCTS.py:92). */
int az_reset_games(az_engine *e);
/* set the root position of every slot (states: [G][words]) and start fresh trees: get_action_visits(env) on a
 * new env.  Also how NBTTT openings (NBTTT/env.py:15-23) are supplied: the caller plays the two random plies. */
int az_set_roots(az_engine *e, const uint64_t *states);
int az_get_roots(az_engine *e, uint64_t *states, int8_t *winner, uint8_t *done);
/* explicit flip bits, consumed one per playout per game: bits[G][n_per_game] (AZ_FLIP_STREAM). */
int az_set_flip_stream(az_engine *e, const uint8_t *bits, int64_t n_per_game);

/* ---- K2 (+K3): search ----------------------------------------------------------------------------------- */
/* n_sims playouts for every slot: MCTS_AZ.get_action_visits' loop (MCTS.py:166-169), evaluator on device.
 * n_sims <= 0 uses cfg.n_playout.  Asynchronous on the engine stream (az_get_* calls synchronise), except for
 * AZ_EVAL_CNN engines of <= 1024 slots with the evaluation cache (<= 32 without) -- first of all the drop-in player's
 * single game -- where playouts that need no network evaluation (terminal node, position in the evaluation cache)
 * are finished in place, a leaf that waits for the network takes its unvisited siblings and its children along into
 * the same K3 launch (their answers go into the cache, where the next playouts find them), and the call returns when
 * the search is complete; the trees are the same bit for bit either way.  AZ_ERR_STATE while rolling self-play
 * (az_selfplay_rounds) is in progress on the engine: az_reset_games / az_set_roots first. */
int az_search(az_engine *e, int32_t n_sims);
/* AZ_EVAL_HOST: one playout for every slot in two halves.  az_select walks to a leaf (MCTS.py:141-144) and
 * returns the position to evaluate (already mirrored when the playout is flipped) and a flag byte
 * (bit0 terminal: evaluation ignored, bit1 flipped).  az_backup takes probs[G][A] over ALL actions and
 * values[G] exactly as CNN.predict returns them (Network.py:85) and performs MCTS.py:147-164. */
int az_select(az_engine *e, uint64_t *leaf_states, uint8_t *flags);
int az_backup(az_engine *e, const float *probs, const float *values);
/* (actions, visits) of the root children in insertion order (MCTS.py:170-171) + their Q and prior.
 * arrays are [G][A]; n_children[G]; root_visits/root_q [G].  Any pointer may be NULL. */
int az_get_root_children(az_engine *e, int32_t *actions, int32_t *visits, float *q, float *prior,
                         int32_t *n_children, int32_t *root_visits, float *root_q);

/* ---- K4: move selection and advancing ----------------------------------------------------------------- */
/* AlphaZeroPlayer.get_action after the search (player.py:104-111): pi[G][A] = softmax(ln N / max(T,1e-8)) as
 * float32, and the move (first-max or sampled, per cfg.move_mode).  temps[G] (or NULL -> temp_all). */
int az_choose_moves(az_engine *e, const double *temps, double temp_all, int32_t *actions, float *pi);
/* play actions[G] (<0: leave the slot alone): env.step + prune_root(action) / fresh root (player.py:112-115).
 * An action >= az_n_actions(game) is AZ_ERR_ARG and nothing is played. */
int az_advance(az_engine *e, const int32_t *actions);

/* ---- self-play collector (game.py:35-58 for G games at once, Pipeline.py:53-63) ------------------------ */
/* Run self-play until `n_games_target` games have finished (slots are refilled with fresh games while
 * started < target).  temp / first_n_steps as game.py:41.  Finished trajectories are appended to the
 * engine's device trajectory buffer; fetch with az_traj_fetch.  Returns number of finished games. */
int64_t az_selfplay(az_engine *e, int64_t n_games_target, double temp, int32_t first_n_steps, int64_t max_steps);
/* one lockstep ply for all live slots (search + choose + advance + record + refill); building block of
 * az_selfplay, exposed for bench.py (a "step").  Asynchronous. */
int az_selfplay_step(az_engine *e, double temp, int32_t first_n_steps, int32_t refill);
/* ROLLING self-play (no counterpart in the reference, which plays one game at a time; it is the batched form of
 * Pipeline.data_collector's loop, Pipeline.py:53-63).  The games do not move in lockstep: every round each live game
 * delivers exactly one NON-terminal leaf to the CNN (a playout that ends on a terminal node is finished on the spot and
 * the game walks again), and the playout that completes a move is followed at once by move selection, env.step,
 * prune_root, trajectory recording and the slot's refill, all inside the tree-step kernel.  Each game still runs its
 * n_playout playouts per move one after the other with its own counter-keyed random streams, so every game's
 * trajectory is the one az_selfplay / az_selfplay_step produce (tests/test_gpu_selfplay.py); leaf batches stay full.
 * az_selfplay_rounds: n_rounds rounds (<= 0: cfg.n_playout), asynchronous; AZ_EVAL_CNN engines only; leaves stay parked
 * between calls; do not mix with az_search / az_selfplay_step without az_reset_games / az_set_roots in between.
 * az_selfplay_rolling: az_selfplay's contract (reset, play until n_games_target games have finished) on rolling rounds. */
int az_selfplay_rounds(az_engine *e, int32_t n_rounds, double temp, int32_t first_n_steps, int32_t refill);
int64_t az_selfplay_rolling(az_engine *e, int64_t n_games_target, double temp, int32_t first_n_steps, int64_t max_calls);
/* total number of games az_selfplay_step's refill may start (0 = unlimited). */
int az_set_game_target(az_engine *e, int64_t n_games_target);
/* park (mask[g] == 0) or resume (mask[g] != 0) slots between calls: a parked slot takes no part in search, ply ends or
 * refill and keeps its game and tree; a slot whose game is over stays idle until a refill restarts it.  In rolling mode a
 * slot parked while it holds a delivered leaf keeps that leaf (and the network's answer to it) untouched: the playout is
 * backed up in the first round after the slot is resumed, so parking never advances ply, tree or trajectory.  No counterpart in
 * the reference (one game at a time); az_selfplay parks surplus slots with it and bench.py staggers the first games so
 * that the timed window sees slots at every game age.  Synchronises. */
int az_set_active(az_engine *e, const uint8_t *mask);
/* counts of finished games / recorded plies so far (synchronises). */
int az_traj_count(az_engine *e, int64_t *n_games, int64_t *n_plies);
/* copy recorded plies (finished games only, in completion order) to the host as flat arrays:
 * states [n][words], pi [n][A] float32, z [n] float32 (mover-relative, game.py:52-55), player [n] int8,
 * action [n] int8, done [n] uint8, game_id [n] uint64.  Returns plies copied; clear!=0 empties the buffer. */
int64_t az_traj_fetch(az_engine *e, int64_t max_plies, uint64_t *states, float *pi, float *z, int8_t *player,
                      int8_t *action, uint8_t *done, uint64_t *game_id, int clear);

/* Transport of finished plies between GPUs (payload of the RCCL all-gather) and into the replay buffer
 * (ReplayBuffer.store, ReplayBuffer.py:68-83).  Packed record = words*8 bytes position | A*4 bytes pi (f32) |
 * i8 z, i8 player, i8 action, u8 done  -> 48 B (Connect4), 48 B (TTT), 72 B (NBTTT). */
int az_traj_record_bytes(int game);
/* pack up to max_plies committed plies into a caller-owned DEVICE buffer; returns plies packed. */
int64_t az_traj_pack_dev(az_engine *e, int64_t max_plies, void *out_dev, int clear);
/* expand n packed records (DEVICE) into ReplayBuffer-shaped DEVICE tensors: state/next_state (n,C,H,W) f32,
 * prob (n,A) f32, value (n,1) f32, winner (n,1) i32, done (n,1) u8.  next_state = position after `action`
 * (game.py:48-49).  Any output may be NULL.  Runs on the current device's default stream, synchronous. */
int az_traj_unpack_dev(int game, int64_t n, const void *packed_dev, float *state_dev, float *prob_dev,
                       float *value_dev, int32_t *winner_dev, float *next_state_dev, uint8_t *done_dev);

/* ---- K3: network --------------------------------------------------------------------------------------- */
/* Upload the Connect4 CNN (Connect4/Network.py:30-55).  blob = every float32 tensor of the reference
 * state_dict concatenated in state_dict order (500106 floats); BatchNorm (eval) is folded here. */
/* A blob that holds a non-finite value is refused (AZ_ERR_ARG), unlike NetworkBase.load (NetworkBase.py:19-24), which
 * carries on with whatever weights it has.  Calling it again on a live engine is the hot reload after a training step
 * (Pipeline.py:87): later searches and forwards use the new weights. */
int az_net_load(az_engine *e, const float *blob, size_t n_floats);
/* CNN.forward + predict (Network.py:60-64,78-85) for n packed Connect4 positions:
 * logp [n][7], logv [n][3], probs [n][7], value [n].  Any output may be NULL. */
int az_net_forward(az_engine *e, int64_t n, const uint64_t *states, float *logp, float *logv, float *probs,
                   float *value);
/* az_net_forward WITHOUT ANY ALLOCATION.  No counterpart in the reference; it is for callers that cannot allocate
 * device memory: tight serving loops (CNN.predict once per move, NetworkPlayer arenas), harnesses that forbid hipMalloc,
 * code around HIP graph capture.  az_net_forward hipMallocs and frees five staging buffers per call; this variant stages
 * the n positions and the four outputs in a fixed device array inside the library (352 KiB per GPU, held until the
 * process ends) and calls no hipMalloc / hipFree.  Same arguments, same kernels, bit-identical outputs.
 * n > AZ_NET_FORWARD_STATIC_MAX is AZ_ERR_ARG.  Every call on a GPU shares that GPU's array: calls that overlap in time
 * on one GPU (two host threads, two engines) race -- serialise them.  The array's address is looked up once per device
 * (az_engine_create) and cached for the life of the process: after a hipDeviceReset it is stale, do not call this again.
 * Devices with an ordinal >= 64 have no staging slot: AZ_ERR_STATE. */
#define AZ_NET_FORWARD_STATIC_MAX 4096
int az_net_forward_static(az_engine *e, int64_t n, const uint64_t *states, float *logp, float *logv, float *probs,
                          float *value);
/* Which kernel evaluates the network from the NEXT az_net_load on (no counterpart in the reference, which has one torch
 * forward): AZ_K3_DEFAULT = the library's product kernel (or, for diagnostic runs, what the environment variable
 * AZ_K3=wino|split|direct names); AZ_K3_WINO = Winograd F(2x2,3x3) on fp32 MFMA; AZ_K3_SPLIT = direct convolution with
 * every fp32 product computed as three fp16 MFMA products on hi/lo split operands, fp32 accumulation, activations rescaled
 * per position by a power of two; AZ_K3_DIRECT = direct convolution on fp32 MFMA (A/B baseline).
 * All three meet the same 1e-4 contract against the reference's fp32 forward.  az_net_get_kernel: the kind in use. */
#define AZ_K3_DEFAULT 0
#define AZ_K3_WINO 1
#define AZ_K3_SPLIT 2
#define AZ_K3_DIRECT 3
int az_net_set_kernel(az_engine *e, int kind);
int az_net_get_kernel(az_engine *e);
/* device-resident forward for bench.py / rocprof: states_dev [n][2], outputs device pointers, async. */
int az_net_forward_dev(az_engine *e, int64_t n, const uint64_t *states_dev, float *probs_dev, float *value_dev);

/* ---- test hooks ------------------------------------------------------------------------------------------ */
/* replace every slot's tree by a root whose children, in this insertion order, have these actions and visit counts
 * (arrays [G][A], n_children[G]), so that az_choose_moves can be checked against recorded outputs of
 * AlphaZeroPlayer.get_action (player.py:101-111) for arbitrary (actions, visits). */
int az_set_root_children(az_engine *e, const int32_t *actions, const int32_t *visits, const int32_t *n_children);
/* TreeNode.PUCT (MCTS.py:47-55) evaluated by the device code path for n independent (Np, n, Q, prior). */
int az_puct_batch(int device, double c_init, int32_t c_base, int64_t n, const int32_t *Np, const int32_t *nv,
                  const float *Q, const float *P, float *score);

#ifdef __cplusplus
}
#endif
#endif /* AZAMD_H */
