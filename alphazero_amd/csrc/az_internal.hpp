// az_internal.hpp -- engine-private declarations shared by the HIP translation units of libazamd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/azamd.h"
#include "az_rules.hpp"

namespace az {

typedef int64_t i64;
typedef int32_t i32;
typedef uint8_t u8;
typedef int8_t i8;
typedef uint16_t u16;

// Node record: 16 bytes, one dwordx4 per child, children of a node contiguous in INSERTION order
// (MCTS.py:10-19 TreeNode; MCTS.py:79-80 select = first max in dict order).
//   x = n_visits (u32)      y = Q (f32 bits)      z = effective prior (f32 bits; 0.75*prior+0.25*noise folded)
//   w = link: bits 0..23 first child (index in this game's arena), 24..27 n_children, 28..31 action
constexpr u32 LINK_FIRST_MASK = 0x00FFFFFFu;
__host__ __device__ inline u32 link_first(u32 l) { return l & LINK_FIRST_MASK; }
__host__ __device__ inline u32 link_nchild(u32 l) { return (l >> 24) & 15u; }
__host__ __device__ inline u32 link_action(u32 l) { return l >> 28; }

enum StatIdx { ST_PLAYOUTS = 0, ST_LEVELS, ST_EVALS, ST_TERMINAL, ST_CHILDREN, ST_GAMES, ST_PLIES, ST_ROLLOUT_PLIES, ST_CACHE_HITS, ST_N };

// Everything a kernel needs, passed by value (all pointers are device memory).
struct Dev {
    i32 game, G, S, A, W, MAXD;
    i32 g0, gn;        // slot range [g0, g0+gn) the split-search kernels of THIS launch cover (a cohort; default: all)
    i64 cap;           // nodes per game arena
    uint4* nodes;      // [G][cap]
    u64* root_state;   // [G][W]
    u32* root;         // [G] index of the root node in the game's arena
    u32* n_alloc;      // [G] nodes in use
    u8* active;        // [G] 1 = slot holds a live (non-terminal) game
    const float2* tab; // [tab_n] {f32(c_init + ln((1+Np+c_base)/c_base)), f32(sqrt(Np))}  (host-built, glibc)
    i32 tab_n;
    const double2* tab64;  // pure MCTS only (else null): {c_init + ln((1+Np+c_base)/c_base), ln(Np)} in float64
    double c_init;
    i32 c_base;
    // leaf evaluation
    i32 evaluator;
    u64 eval_seed;
    // flips
    i32 flip_mode;
    const u8* flip_bits;  // [G][flip_n]
    i64 flip_n;
    u32* flip_pos;        // [G]
    u64* sim_ctr;         // [G] playouts ever run in this slot's current game (RNG counter)
    i32* move_sims;       // [G] rolling self-play: playouts done for the move being searched
    u64* game_id;         // [G]
    u64 gid_stride;       // global game id of the k-th game this engine starts = game_id_base + gid_stride * k
    u64 rng_seed;
    float noise_alpha;    // <=0: off
    // split-phase buffers (select -> evaluate -> backup)
    u32* path_idx;        // [G][MAXD]
    u32* path_N;          // [G][MAXD]
    float* path_Q;        // [G][MAXD]
    i32* leaf_depth;      // [G]
    u32* leaf_flags;      // [G] bit0 terminal, bit1 flipped, bit2 symmetric, bit3 valid (slot took part)
    float* leaf_tvalue;   // [G] terminal value
    u64* leaf_state;      // [G][W] position handed to the evaluator (mirrored when flipped)
    float* eval_probs;    // [G][A]
    float* eval_value;    // [G]
    u32* eval_list;       // [G] compacted list of slots whose leaf needs the network (a cohort's list starts at g0)
    u32* eval_count;      // [1] per cohort; k_select counts up, k_backup clears it for the next round
    // speculative evaluation (az_search of a handful of games, az_tree.hip "Speculation"): spec_rows extra rows per slot behind the
    // G slot rows of leaf_state / eval_probs / eval_value / eval_list (row G + g * spec_rows + r), 0 = off
    i32 spec_rows;
    u8* spec_mark;        // [G][spec_rows] != 0: the row holds a position whose evaluation goes into the cache at the next tree step
    u8 spec_tag;          // what a mark is set to: changes with every az_net_load (a row marked under other weights is dropped)
    // status
    i8* winner;           // [G]
    u8* done;             // [G]
    i32* ply;             // [G]
    unsigned long long* stats;  // [ST_N]
    u32* err;             // [1] bit0 arena overflow, bit1 puct table overflow
    // evaluation cache (AZ_EVAL_CNN; az_tree.hip "Evaluation cache"): direct-mapped table of 64-byte entries shared by all
    // the games of the engine, or null
    uint4* cache;         // [cache_mask + 1][4]
    u64 cache_mask;       // entries - 1 (a power of two)
    u64 cache_epoch;      // folded into every entry's tags: changes with every az_net_load, which empties the table at once
};

struct CnnDev;  // az_cnn.hip

void set_error(const std::string& s);
#define AZ_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t _e = (call);                                                              \
        if (_e != hipSuccess) {                                                              \
            az::set_error(std::string(#call) + ": " + hipGetErrorString(_e));                \
            (void)hipGetLastError(); /* reported here: not again by a later launch check */  \
            return AZ_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

// launchers implemented in the .hip files
int launch_env_step(int game, i64 n, const u64* in, const i32* act, u64* out, u8* placed, i8* winner, u8* done,
                    u16* legal, hipStream_t st);
int launch_env_status(int game, i64 n, const u64* in, i8* winner, u8* done, u16* legal, u8* full, hipStream_t st);
int launch_env_planes(int game, i64 n, const u64* in, float* planes, hipStream_t st);
int launch_env_flip(int game, i64 n, const u64* in, u64* out, i8* fa, hipStream_t st);
int launch_puct(double c_init, i32 c_base, i64 n, const i32* Np, const i32* nv, const float* Q, const float* P,
                const float2* tab, i32 tab_n, float* score, hipStream_t st);
int launch_search_fused(const Dev& d, int n_sims, hipStream_t st);
int launch_search_pure(const Dev& d, int n_sims, hipStream_t st);
int launch_select(const Dev& d, hipStream_t st, bool clear_count);
int launch_backup(const Dev& d, hipStream_t st);
int launch_backup_select(const Dev& d, u32* stale_count, hipStream_t st);
int launch_root_children(const Dev& d, i32* actions, i32* visits, float* q, float* prior, i32* nch, i32* rootN,
                         float* rootQ, hipStream_t st);
int launch_reset_trees(const Dev& d, const u8* only /* [G] or null */, hipStream_t st);

}  // namespace az
