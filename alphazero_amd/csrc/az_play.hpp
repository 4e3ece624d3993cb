// az_play.hpp -- device-side buffers of the self-play collector (K4) shared by az_play.hip and az_engine.hip.
#pragma once
#include "az_internal.hpp"

namespace az {

struct Play {
    // per-slot staging of the game in progress: [G][MAX_PLIES]
    u64* tj_state;   // [..][W]
    float* tj_pi;    // [..][A]
    i8* tj_action;
    i8* tj_player;
    // committed (finished) games, in completion order: [out_cap] plies
    i64 out_cap;
    u64* out_state;  // [out_cap][W]
    float* out_pi;   // [out_cap][A]
    float* out_z;
    i8* out_player;
    i8* out_action;
    u8* out_done;
    u64* out_gid;
    unsigned long long* out_count;      // plies committed
    unsigned long long* out_games;      // games committed
    unsigned long long* games_started;  // refill counter
};

int launch_choose(const Dev& d, const Play& p, const double* temps, double temp_all, int first_n, int move_mode,
                  i32* actions, float* pi, hipStream_t st);
int launch_advance(const Dev& d, const Play& p, const i32* actions, const float* pi, int reuse, int record,
                   hipStream_t st);
int launch_refill(const Dev& d, const Play& p, unsigned long long target, u64 gid_base, hipStream_t st);
// choose + advance (recording) + refill (target != 0) of every slot in one launch
int launch_ply_end(const Dev& d, const Play& p, double temp_all, int first_n, int move_mode, int reuse,
                   unsigned long long target, u64 gid_base, hipStream_t st);
int launch_init_slots(const Dev& d, int use_openings, u64 gid_base, hipStream_t st);
// rolling self-play tree step (az_tree.hip): mode 2 = the very first selection, 3 = finish round i (with the ply ends that
// fall into it) and select round i+1; stale_count = the leaf counter the evaluator has just consumed (cleared here).
// search_cnt != null: az_search on the rolling step (S = playouts of this search, no ply ends): [0] counts the games that take
// part (mode 2), [1] the games whose S playouts are complete
int launch_roll_step(const Dev& d, int mode, u32* stale_count, const Play& p, int S, double temp, int first_n, int move_mode,
                     int reuse, unsigned long long target, u64 gid_base, hipStream_t st, u32* search_cnt = nullptr);
int launch_traj_pack(int game, const Play& p, i64 n, u8* out, hipStream_t st);
int launch_traj_unpack(int game, i64 n, const u8* recs, float* state, float* prob, float* value, i32* winner,
                       float* next_state, u8* done, hipStream_t st);

}  // namespace az
