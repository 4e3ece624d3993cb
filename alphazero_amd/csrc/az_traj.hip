// az_traj.hip -- trajectory transport: pack finished plies into fixed-size records (the payload of the RCCL
// all-gather) and expand records into ReplayBuffer-shaped tensors on the consumer.
//
// Reference sink: ReplayBuffer.store(state, prob, value, winner, next_state, done) (ReplayBuffer.py:68-83) fed by
// Pipeline.data_collector (Pipeline.py:58-62) with the 6-tuples of Game.start_self_play (game.py:58).
//
// Packed record (little endian): W x u64 position before the move | A x f32 pi | i8 z, i8 player, i8 action, u8 done
//   Connect4 48 B, TTT 48 B, NBTTT 72 B   (vs 1 045 B for the reference's unpacked Connect4 tuple)
#include "az_internal.hpp"
#include "az_play.hpp"

namespace az {

template <class G>
__global__ __launch_bounds__(256) void k_traj_pack(Play p, i64 n, u8* out) {
    constexpr int REC = G::W * 8 + G::A * 4 + 4;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        u8* r = out + i * REC;
        u64* rs = reinterpret_cast<u64*>(r);
        for (int k = 0; k < G::W; k++) rs[k] = p.out_state[i * G::W + k];
        float* rp = reinterpret_cast<float*>(r + G::W * 8);
        for (int k = 0; k < G::A; k++) rp[k] = p.out_pi[i * G::A + k];
        i8* rm = reinterpret_cast<i8*>(r + G::W * 8 + G::A * 4);
        rm[0] = (i8)p.out_z[i];
        rm[1] = p.out_player[i];
        rm[2] = p.out_action[i];
        rm[3] = (i8)p.out_done[i];
    }
}

template <class G>
__device__ inline float plane_value(const typename G::State& s, int r) {
    if constexpr (G::GAME == AZ_GAME_C4) {  // env_cython.pyx:143-149
        int ch = r / 42, q = r % 42, row = q / 7, col = q % 7, bit = col * 7 + (5 - row);
        if (ch == 0) return (float)((C4::xs(s) >> bit) & 1);
        if (ch == 1) return (float)((C4::os(s) >> bit) & 1);
        return (float)C4::turn(s);
    } else if constexpr (G::GAME == AZ_GAME_TTT) {  // TTT/utils.py:5-11
        int ch = r / 9, q = r % 9;
        if (ch == 0) return (float)((TTT::xs(s) >> q) & 1);
        if (ch == 1) return (float)((TTT::os(s) >> q) & 1);
        return (float)TTT::turn(s);
    } else {  // NBTTT/utils.py:258-268 with 0-based current-board planes
        int ch = r / 9, q = r % 9;
        if (ch < 9) return (float)((NBTTT::sub(s, ch) >> q) & 1);
        if (ch < 18) return (float)((NBTTT::sub(s, ch - 9) >> (9 + q)) & 1);
        if (ch < 27) return (ch - 18 == NBTTT::curr(s)) ? 1.f : 0.f;
        return (float)NBTTT::turn(s);
    }
}

template <class G>
__global__ __launch_bounds__(256) void k_traj_unpack(i64 n, const u8* __restrict__ recs, float* __restrict__ state,
                                                     float* __restrict__ prob, float* __restrict__ value,
                                                     i32* __restrict__ winner, float* __restrict__ next_state,
                                                     u8* __restrict__ done) {
    constexpr int REC = G::W * 8 + G::A * 4 + 4;
    constexpr int PL = G::GAME == AZ_GAME_C4 ? 126 : (G::GAME == AZ_GAME_TTT ? 27 : 252);
    const i64 total = n * PL;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
        i64 i = t / PL;
        int r = (int)(t - i * PL);
        const u8* rec = recs + i * REC;
        typename G::State s;
        for (int k = 0; k < G::W; k++) s.w[k] = reinterpret_cast<const u64*>(rec)[k];
        const i8* rm = reinterpret_cast<const i8*>(rec + G::W * 8 + G::A * 4);
        if (state) state[t] = plane_value<G>(s, r);
        if (next_state) {
            typename G::State s2 = s;
            G::step(s2, (int)rm[2]);  // game.py:48-49
            next_state[t] = plane_value<G>(s2, r);
        }
        if (r < G::A && prob) prob[i * G::A + r] = reinterpret_cast<const float*>(rec + G::W * 8)[r];
        if (r == 0) {
            if (value) value[i] = (float)rm[0];
            if (winner) winner[i] = (i32)rm[0];
            if (done) done[i] = (u8)rm[3];
        }
    }
}

static inline int grid_for(i64 n) {
    i64 b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

#define AZ_DISPATCH_GAME(game, ...)                                 \
    switch (game) {                                                 \
        case AZ_GAME_C4: { using GT = C4; __VA_ARGS__; } break;     \
        case AZ_GAME_TTT: { using GT = TTT; __VA_ARGS__; } break;   \
        case AZ_GAME_NBTTT: { using GT = NBTTT; __VA_ARGS__; } break; \
        default: set_error("unknown game"); return AZ_ERR_ARG;      \
    }

int launch_traj_pack(int game, const Play& p, i64 n, u8* out, hipStream_t st) {
    if (n <= 0) return AZ_OK;
    AZ_DISPATCH_GAME(game, k_traj_pack<GT><<<dim3(grid_for(n)), dim3(256), 0, st>>>(p, n, out));
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

int launch_traj_unpack(int game, i64 n, const u8* recs, float* state, float* prob, float* value, i32* winner,
                       float* next_state, u8* done, hipStream_t st) {
    if (n <= 0) return AZ_OK;
    AZ_DISPATCH_GAME(game, {
        constexpr int PL = GT::GAME == AZ_GAME_C4 ? 126 : (GT::GAME == AZ_GAME_TTT ? 27 : 252);
        k_traj_unpack<GT><<<dim3(grid_for(n * PL)), dim3(256), 0, st>>>(n, recs, state, prob, value, winner, next_state,
                                                                       done);
    });
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

}  // namespace az
