// az_play.hip -- K4: move selection, advancing the games, trajectory recording, slot refill (gfx950).
//
// Replaces, for G games at once:
//   AlphaZeroPlayer.get_action  player.py:101-116  -> k_choose: pi = softmax(ln N / max(T,1e-8)) in float64 -> float32,
//                                                     T == 0: uniform over the maxima; deterministic or sampled move
//   utils.softmax               utils.py:55-57
//   MCTS.prune_root             MCTS.py:130-135    -> k_advance: re-root on the chosen child (stats kept) or fresh root
//   Game.start_self_play        game.py:35-58      -> k_advance records (state, pi, mover) per ply; on game end the
//                                                     mover-relative outcome z is filled and the game is committed to
//                                                     the trajectory output arrays; k_refill restarts finished slots
// One lane per game: these kernels run once per ply (the tree kernels run S times per ply).
#include "az_internal.hpp"
#include "az_play.hpp"
#include "az_play_dev.hpp"

namespace az {

template <class G>
__global__ __launch_bounds__(256) void k_choose(Dev d, Play p, const double* temps, double temp_all, int first_n,
                                               int move_mode, i32* actions, float* pi) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    actions[g] = -1;
    if (!d.active[g]) {
        for (int a = 0; a < G::A; a++) pi[(i64)g * G::A + a] = 0.f;
        return;
    }
    double T = temps ? temps[g] : temp_all;
    if (!temps && first_n >= 0) T = (d.ply[g] >= first_n) ? 1e-3 : temp_all;  // game.py:41
    actions[g] = choose_game<G>(d, g, T, move_mode, pi + (i64)g * G::A, 1);
}

// env.step(action) + prune_root(action)/reset_player() + trajectory bookkeeping for every slot.
template <class G>
__global__ __launch_bounds__(256) void k_advance(Dev d, Play p, const i32* actions, const float* pi, int reuse,
                                                int record) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    if (!d.active[g]) return;
    int a = actions[g];
    if (a < 0) return;
    advance_game<G>(d, p, g, a, pi + (i64)g * G::A, 1, reuse, record);
}

// Restart finished slots with a fresh game while fewer than `target` games have been started.
template <class G>
__global__ __launch_bounds__(256) void k_refill(Dev d, Play p, unsigned long long target, u64 gid_base) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    refill_game<G>(d, p, g, target, gid_base);
}

// One whole ply end per slot: k_choose + k_advance(record) + k_refill in a single launch (az_selfplay_step).
template <class G>
__global__ __launch_bounds__(256) void k_ply_end(Dev d, Play p, double temp_all, int first_n, int move_mode, int reuse,
                                                unsigned long long target, u64 gid_base) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    if (d.active[g]) {
        float pi[G::A];
        const double T = (first_n >= 0 && d.ply[g] >= first_n) ? 1e-3 : temp_all;  // game.py:41
        const int a = choose_game<G>(d, g, T, move_mode, pi, 1);
        if (a >= 0) advance_game<G>(d, p, g, a, pi, 1, reuse, 1);
    }
    if (target) refill_game<G>(d, p, g, target, gid_base);
}

// After az_set_roots / az_reset_games: derive status of every slot from its root position.
template <class G>
__global__ __launch_bounds__(256) void k_init_slots(Dev d, int use_openings, u64 gid_base) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    typename G::State s;
    u64 gid = gid_base + d.gid_stride * (u64)g;
    if (use_openings) {
        s = opening_state<G>(d, gid);
        for (int i = 0; i < G::W; i++) d.root_state[(i64)g * G::W + i] = s.w[i];
    } else {
        for (int i = 0; i < G::W; i++) s.w[i] = d.root_state[(i64)g * G::W + i];
    }
    int w = G::winner(s);
    bool dn = (w != 0) || G::full(s);
    d.winner[g] = (i8)w;
    d.done[g] = dn ? 1 : 0;
    d.active[g] = dn ? 0 : 1;
    d.ply[g] = 0;
    d.sim_ctr[g] = 0;
    d.move_sims[g] = 0;
    d.flip_pos[g] = 0;
    d.game_id[g] = gid;
    d.nodes[(i64)g * d.cap] = make_uint4(0u, 0u, __float_as_uint(1.0f), 0u);
    d.root[g] = 0u;
    d.n_alloc[g] = 1u;
    d.leaf_flags[g] = 0u;
}

#define AZ_DISPATCH_GAME(game, ...)                                 \
    switch (game) {                                                 \
        case AZ_GAME_C4: { using GT = C4; __VA_ARGS__; } break;     \
        case AZ_GAME_TTT: { using GT = TTT; __VA_ARGS__; } break;   \
        case AZ_GAME_NBTTT: { using GT = NBTTT; __VA_ARGS__; } break; \
        default: set_error("unknown game"); return AZ_ERR_ARG;      \
    }

int launch_choose(const Dev& d, const Play& p, const double* temps, double temp_all, int first_n, int move_mode,
                  i32* actions, float* pi, hipStream_t st) {
    AZ_DISPATCH_GAME(d.game, k_choose<GT><<<dim3((d.G + 255) / 256), dim3(256), 0, st>>>(d, p, temps, temp_all, first_n, move_mode, actions, pi));
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

int launch_advance(const Dev& d, const Play& p, const i32* actions, const float* pi, int reuse, int record,
                   hipStream_t st) {
    AZ_DISPATCH_GAME(d.game, k_advance<GT><<<dim3((d.G + 255) / 256), dim3(256), 0, st>>>(d, p, actions, pi, reuse, record));
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

int launch_ply_end(const Dev& d, const Play& p, double temp_all, int first_n, int move_mode, int reuse,
                   unsigned long long target, u64 gid_base, hipStream_t st) {
    AZ_DISPATCH_GAME(d.game, k_ply_end<GT><<<dim3((d.G + 255) / 256), dim3(256), 0, st>>>(d, p, temp_all, first_n, move_mode, reuse, target, gid_base));
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

int launch_refill(const Dev& d, const Play& p, unsigned long long target, u64 gid_base, hipStream_t st) {
    AZ_DISPATCH_GAME(d.game, k_refill<GT><<<dim3((d.G + 255) / 256), dim3(256), 0, st>>>(d, p, target, gid_base));
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

int launch_init_slots(const Dev& d, int use_openings, u64 gid_base, hipStream_t st) {
    AZ_DISPATCH_GAME(d.game, k_init_slots<GT><<<dim3((d.G + 255) / 256), dim3(256), 0, st>>>(d, use_openings, gid_base));
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

}  // namespace az
