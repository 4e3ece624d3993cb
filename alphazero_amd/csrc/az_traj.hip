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

// Connect4 streaming form.  A ply's planes are 126 floats = 63 float2: one lane per float2 of `state` and of `next_state`,
// 4 plies per 256-lane workgroup and trip (252 lanes busy), every lane's (ply-in-group, pair) fixed for the whole
// kernel, no per-element division: the (channel, bit) of plane element r comes from a 126-byte table in LDS.  8-byte
// non-temporal stores, fully coalesced (a ply's rows are 504 B, so rows are 8-byte aligned wherever the ring pointer
// of the ReplayBuffer stands).  The generic kernel above paid a 64-bit division, a record load and an env.step per
// FLOAT: 1.6 TB/s; this one is measured in DESIGN.md section 5.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef AZ_UNPACK_NT
#define AZ_UNPACK_NT 1
#endif
__device__ inline void unpack_store(f32x2* p, f32x2 v) {
#if AZ_UNPACK_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__global__ __launch_bounds__(256) void k_traj_unpack_c4(i64 n, const u8* __restrict__ recs, float* __restrict__ state,
                                                        float* __restrict__ prob, float* __restrict__ value,
                                                        i32* __restrict__ winner, float* __restrict__ next_state,
                                                        u8* __restrict__ done) {
    constexpr int REC = 48, PL = 126;
    __shared__ u8 code[128];  // element r of a ply: bits 0-5 bitboard bit, bit 6 = O plane, bit 7 = turn plane
    const int tid = threadIdx.x;
    if (tid < PL) {
        const int ch = tid / 42, q = tid % 42, row = q / 7, col = q % 7;
        code[tid] = (u8)(ch == 2 ? 128 : (col * 7 + (5 - row)) | (ch == 1 ? 64 : 0));
    }
    __syncthreads();
    const int sub = tid / 63, pr = tid - sub * 63;  // ply within the group of 4, float2 within the ply
    if (sub >= 4) return;
    const u32 c0 = code[2 * pr], c1 = code[2 * pr + 1];
    // branch-free: which word, which bit, or the turn plane -- all fixed per lane for the whole kernel
    auto plane = [](u64 xs, u64 os, float turn, u32 c) -> float {
        const u64 w = (c & 64u) ? os : xs;
        const float bit = (float)(u32)((w >> (c & 63u)) & 1ull);
        return (c & 128u) ? turn : bit;
    };
    const i64 groups = (n + 3) / 4;
    for (i64 g = blockIdx.x; g < groups; g += gridDim.x) {
        const i64 i = g * 4 + sub;
        if (i >= n) continue;
        const u8* rec = recs + i * REC;
        C4::State s;
        s.w[0] = reinterpret_cast<const u64*>(rec)[0];
        s.w[1] = reinterpret_cast<const u64*>(rec)[1];
        const u32 meta = *reinterpret_cast<const u32*>(rec + 44);  // i8 z | i8 player | i8 action | u8 done
        const int z = (int)(i8)(meta & 0xFFu);
        if (state) {
            const u64 xs = C4::xs(s), os = C4::os(s);
            const float tn = (float)C4::turn(s);
            f32x2 v;
            v.x = plane(xs, os, tn, c0);
            v.y = plane(xs, os, tn, c1);
            unpack_store(reinterpret_cast<f32x2*>(state + i * PL) + pr, v);
        }
        if (next_state) {
            C4::State s2 = s;
            C4::step(s2, (int)(i8)((meta >> 16) & 0xFFu));  // game.py:48-49
            const u64 xs = C4::xs(s2), os = C4::os(s2);
            const float tn = (float)C4::turn(s2);
            f32x2 v;
            v.x = plane(xs, os, tn, c0);
            v.y = plane(xs, os, tn, c1);
            unpack_store(reinterpret_cast<f32x2*>(next_state + i * PL) + pr, v);
        }
        if (pr < 7) {
            if (prob) prob[i * 7 + pr] = reinterpret_cast<const float*>(rec + 16)[pr];
        } else if (pr == 7) {
            if (value) value[i] = (float)z;
        } else if (pr == 8) {
            if (winner) winner[i] = (i32)z;
        } else if (pr == 9) {
            if (done) done[i] = (u8)(meta >> 24);
        }
    }
}

// Whole-line form for the bulk of a batch.  32 consecutive plies are 16 128 B of planes = 126 complete 128-byte lines
// per tensor (and 7 lines of pi, 1 of value, 1 of winner), so when the first of them starts on a line boundary every
// wave stores whole lines with 16-byte non-temporal stores -- no line is written in two pieces at two times, which is
// what held the float2 form at half the rate of a plain fill.  Lanes 0..31 load the 32 records once, play the move
// (game.py:48-49) and park {X, O | turn} of the position before and after in LDS; every lane then owns the same four
// float4 of the chunk on every trip (which ply, which (word, bit) per element, whether it straddles into the next ply:
// worked out once, before the loop).  The launcher peels the plies before the first line boundary and the ragged tail
// off to k_traj_unpack_c4.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int UC = 32;                   // plies per chunk
constexpr int UC_F4 = UC * 126 / 4;      // 1 008 float4 per tensor and chunk
__device__ inline float plane_of(ulonglong2 rec, u32 c) {
    const u64 w = (c & 64u) ? rec.y : rec.x;                           // bits 0..48 only: the turn flag (bit 63) never shows
    const float bit = (float)(u32)((w >> (c & 63u)) & 1ull);
    const float turn = (rec.y >> 63) ? -1.f : 1.f;
    return (c & 128u) ? turn : bit;
}
__global__ __launch_bounds__(256) void k_traj_unpack_c4_lines(i64 n_chunks, const u8* __restrict__ recs,
                                                              float* __restrict__ state, float* __restrict__ prob,
                                                              float* __restrict__ value, i32* __restrict__ winner,
                                                              float* __restrict__ next_state, u8* __restrict__ done) {
    constexpr int REC = 48, PL = 126;
    __shared__ u8 code[128];
    __shared__ ulonglong2 cur[UC + 1], nxt[UC + 1];  // [UC]: never selected (the chunk's last float4 does not straddle)
    const int tid = threadIdx.x;
    if (tid < PL) {
        const int ch = tid / 42, q = tid % 42, row = q / 7, col = q % 7;
        code[tid] = (u8)(ch == 2 ? 128 : (col * 7 + (5 - row)) | (ch == 1 ? 64 : 0));
    }
    if (tid == 0) cur[UC] = nxt[UC] = make_ulonglong2(0ull, 0ull);
    __syncthreads();
    int pa[4];       // ply (within the chunk) of the float4's first element
    u32 cd[4];       // the four elements' codes, one byte each
    bool straddle[4];  // elements 2 and 3 belong to ply pa + 1 (126 = 4 * 31 + 2: only an element pair can cross)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int f = tid + 256 * k, e = 4 * f;
        pa[k] = (f < UC_F4) ? e / PL : 0;
        const int r = (f < UC_F4) ? e - PL * pa[k] : 0;
        straddle[k] = (r == PL - 2);
        cd[k] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int rr = r + j;
            if (rr >= PL) rr -= PL;
            cd[k] |= (u32)code[rr] << (8 * j);
        }
    }
    for (i64 c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const i64 base = c * UC;
        if (tid < UC) {
            const i64 i = base + tid;
            const u8* rec = recs + i * REC;
            C4::State s;
            s.w[0] = reinterpret_cast<const u64*>(rec)[0];
            s.w[1] = reinterpret_cast<const u64*>(rec)[1];
            const u32 meta = *reinterpret_cast<const u32*>(rec + 44);  // i8 z | i8 player | i8 action | u8 done
            const int z = (int)(i8)(meta & 0xFFu);
            cur[tid] = make_ulonglong2(s.w[0], s.w[1]);
            C4::step(s, (int)(i8)((meta >> 16) & 0xFFu));
            nxt[tid] = make_ulonglong2(s.w[0], s.w[1]);
            if (value) value[i] = (float)z;
            if (winner) winner[i] = (i32)z;
            if (done) done[i] = (u8)(meta >> 24);
        }
        if (prob && tid < UC * 7) {  // 224 consecutive floats
            const int pl = tid / 7, a = tid - 7 * pl;
            prob[base * 7 + tid] = reinterpret_cast<const float*>(recs + (base + pl) * REC + 16)[a];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int f = tid + 256 * k;
            if (f < UC_F4) {
                const i64 o = base * PL / 4 + f;  // base is a multiple of 32 plies: 32 * 126 / 4 float4 per chunk
                if (state) {
                    const ulonglong2 a = cur[pa[k]], b = cur[pa[k] + 1];
                    const ulonglong2 hi = straddle[k] ? b : a;
                    f32x4 v;
                    v.x = plane_of(a, cd[k] & 0xFFu);
                    v.y = plane_of(a, (cd[k] >> 8) & 0xFFu);
                    v.z = plane_of(hi, (cd[k] >> 16) & 0xFFu);
                    v.w = plane_of(hi, cd[k] >> 24);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(state) + o);
                }
                if (next_state) {
                    const ulonglong2 a = nxt[pa[k]], b = nxt[pa[k] + 1];
                    const ulonglong2 hi = straddle[k] ? b : a;
                    f32x4 v;
                    v.x = plane_of(a, cd[k] & 0xFFu);
                    v.y = plane_of(a, (cd[k] >> 8) & 0xFFu);
                    v.z = plane_of(hi, (cd[k] >> 16) & 0xFFu);
                    v.w = plane_of(hi, cd[k] >> 24);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(next_state) + o);
                }
            }
        }
        __syncthreads();  // the next trip rewrites cur / nxt
    }
}

// The same whole-line scheme for the other two plane formats (TTT: 27 floats per ply, 32 plies = 27 lines; NBTTT: 252
// floats, 8 plies = 63 lines).  Every plane element is one 64-bit word of the position plus a rule: a bit of it
// (stones), its top bit (turn plane), or its low nibble compared with a constant (NBTTT "current board" planes); the
// words of the chunk's positions before and after the move sit in LDS and each lane reads the word each of its
// elements needs (which ply, which word, which rule: worked out once, before the loop).
template <class G>
struct UnpackGeo {
    static constexpr int PL = G::GAME == AZ_GAME_TTT ? 27 : 252;  // floats per ply
    static constexpr int CH = G::GAME == AZ_GAME_TTT ? 32 : 8;    // plies per chunk: CH * PL * 4 bytes = whole lines
    static constexpr int NF4 = CH * PL / 4;
    static constexpr int K = (NF4 + 255) / 256;
};
template <class G>
__device__ inline u32 plane_rule(int r) {  // bits 0-5 bit, 6-7 word, 8-9 kind (0 bit, 1 turn, 2 nibble == k), 10-13 k
    const int ch = r / 9, q = r % 9;
    if constexpr (G::GAME == AZ_GAME_TTT) {  // TTT/utils.py:5-11
        if (ch == 0) return (u32)q;
        if (ch == 1) return (u32)(16 + q);
        return 1u << 8;
    } else {  // NBTTT/utils.py:258-268 with 0-based current-board planes
        if (ch < 9) return (u32)(18 * (ch % 3) + q) | ((u32)(ch / 3) << 6);
        if (ch < 18) return (u32)(18 * ((ch - 9) % 3) + 9 + q) | ((u32)((ch - 9) / 3) << 6);
        if (ch < 27) return (2u << 8) | (3u << 6) | ((u32)(ch - 18) << 10);
        return (1u << 8) | (3u << 6);
    }
}
__device__ inline float plane_by_rule(u64 w, u32 rule) {
    const u32 kind = (rule >> 8) & 3u;
    const float bit = (float)(u32)((w >> (rule & 63u)) & 1ull);
    const float turn = (w >> 63) ? -1.f : 1.f;
    const float is_k = ((u32)(w & 15ull) == ((rule >> 10) & 15u)) ? 1.f : 0.f;
    return kind == 0 ? bit : (kind == 1 ? turn : is_k);
}
template <class G>
__global__ __launch_bounds__(256) void k_traj_unpack_lines(i64 n_chunks, const u8* __restrict__ recs,
                                                           float* __restrict__ state, float* __restrict__ prob,
                                                           float* __restrict__ value, i32* __restrict__ winner,
                                                           float* __restrict__ next_state, u8* __restrict__ done) {
    using Geo = UnpackGeo<G>;
    constexpr int REC = G::W * 8 + G::A * 4 + 4, PL = Geo::PL, CH = Geo::CH, W = G::W;
    __shared__ u64 cur[CH * W], nxt[CH * W];
    const int tid = threadIdx.x;
    u32 slot[Geo::K][4];  // per element: LDS word index (bits 16+) | rule
#pragma unroll
    for (int k = 0; k < Geo::K; k++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int e = 4 * (tid + 256 * k) + j;
            const int pl = e < CH * PL ? e / PL : 0, r = e < CH * PL ? e - PL * pl : 0;
            const u32 rule = plane_rule<G>(r);
            slot[k][j] = rule | ((u32)(pl * W + (int)((rule >> 6) & 3u)) << 16);
        }
    }
    for (i64 c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const i64 base = c * CH;
        if (tid < CH) {
            const i64 i = base + tid;
            const u8* rec = recs + i * REC;
            typename G::State s;
#pragma unroll
            for (int k = 0; k < W; k++) {
                s.w[k] = reinterpret_cast<const u64*>(rec)[k];
                cur[tid * W + k] = s.w[k];
            }
            const u32 meta = *reinterpret_cast<const u32*>(rec + W * 8 + G::A * 4);  // i8 z | player | action | u8 done
            const int z = (int)(i8)(meta & 0xFFu);
            G::step(s, (int)(i8)((meta >> 16) & 0xFFu));  // game.py:48-49
#pragma unroll
            for (int k = 0; k < W; k++) nxt[tid * W + k] = s.w[k];
            if (value) value[i] = (float)z;
            if (winner) winner[i] = (i32)z;
            if (done) done[i] = (u8)(meta >> 24);
        }
        if (prob) {
            for (int t = tid; t < CH * G::A; t += 256) {
                const int pl = t / G::A, a = t - G::A * pl;
                prob[base * G::A + t] = reinterpret_cast<const float*>(recs + (base + pl) * REC + W * 8)[a];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < Geo::K; k++) {
            const int f = tid + 256 * k;
            if (f < Geo::NF4) {
                const i64 o = base * PL / 4 + f;  // base is a multiple of CH plies and CH * PL of 4
                if (state) {
                    f32x4 v;
                    v.x = plane_by_rule(cur[slot[k][0] >> 16], slot[k][0]);
                    v.y = plane_by_rule(cur[slot[k][1] >> 16], slot[k][1]);
                    v.z = plane_by_rule(cur[slot[k][2] >> 16], slot[k][2]);
                    v.w = plane_by_rule(cur[slot[k][3] >> 16], slot[k][3]);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(state) + o);
                }
                if (next_state) {
                    f32x4 v;
                    v.x = plane_by_rule(nxt[slot[k][0] >> 16], slot[k][0]);
                    v.y = plane_by_rule(nxt[slot[k][1] >> 16], slot[k][1]);
                    v.z = plane_by_rule(nxt[slot[k][2] >> 16], slot[k][2]);
                    v.w = plane_by_rule(nxt[slot[k][3] >> 16], slot[k][3]);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(next_state) + o);
                }
            }
        }
        __syncthreads();  // the next trip rewrites cur / nxt
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

// TTT / NBTTT: plies before the first 128-byte line of the plane tensors and the ragged tail through the per-float
// kernel, whole chunks through k_traj_unpack_lines.
template <class G>
static int unpack_lines_or_ragged(i64 n, const u8* recs, float* state, float* prob, float* value, i32* winner,
                                  float* next_state, u8* done, hipStream_t st) {
    using Geo = UnpackGeo<G>;
    constexpr int REC = G::W * 8 + G::A * 4 + 4, PL = Geo::PL;
    auto ragged = [&](i64 lo, i64 m) {
        if (m <= 0) return;
        k_traj_unpack<G><<<dim3(grid_for(m * PL)), dim3(256), 0, st>>>(
            m, recs + lo * REC, state ? state + lo * PL : nullptr, prob ? prob + lo * G::A : nullptr,
            value ? value + lo : nullptr, winner ? winner + lo : nullptr, next_state ? next_state + lo * PL : nullptr,
            done ? done + lo : nullptr);
    };
    const uintptr_t sa = (uintptr_t)(state ? state : next_state), sb = (uintptr_t)(next_state ? next_state : state);
    i64 head = n;
    if (sa && (sa & 127u) == (sb & 127u) && ((uintptr_t)recs & 7u) == 0)
        for (i64 h = 0; h < 128 && h <= n; h++)
            if (((sa + (uintptr_t)h * PL * 4) & 127u) == 0) {
                head = h;
                break;
            }
    const i64 chunks = (n - head) / Geo::CH;
    ragged(0, head);
    if (chunks > 0) {
        const int grid = (int)(chunks < 4096 ? chunks : 4096);
        k_traj_unpack_lines<G><<<dim3(grid), dim3(256), 0, st>>>(
            chunks, recs + head * REC, state ? state + head * PL : nullptr, prob ? prob + head * G::A : nullptr,
            value ? value + head : nullptr, winner ? winner + head : nullptr,
            next_state ? next_state + head * PL : nullptr, done ? done + head : nullptr);
    }
    ragged(head + chunks * Geo::CH, n - head - chunks * Geo::CH);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

int launch_traj_unpack(int game, i64 n, const u8* recs, float* state, float* prob, float* value, i32* winner,
                       float* next_state, u8* done, hipStream_t st) {
    if (n <= 0) return AZ_OK;
    if (game == AZ_GAME_C4 && (((uintptr_t)state | (uintptr_t)next_state | (uintptr_t)recs) & 7u) == 0) {
        auto ragged = [&](i64 lo, i64 m) {  // plies [lo, lo + m) through the float2 kernel
            if (m <= 0) return;
            const i64 groups = (m + 3) / 4;
            const int grid = (int)(groups < 8192 ? groups : 8192);
            k_traj_unpack_c4<<<dim3(grid), dim3(256), 0, st>>>(
                m, recs + lo * 48, state ? state + lo * 126 : nullptr, prob ? prob + lo * 7 : nullptr,
                value ? value + lo : nullptr, winner ? winner + lo : nullptr, next_state ? next_state + lo * 126 : nullptr,
                done ? done + lo : nullptr);
        };
        // plies until the plane rows start on a 128-byte line: a row is 504 B = 128 * 4 - 8, so each ply moves the start
        // back by 8 bytes -- `head` plies with head = (address mod 128) / 8.  Both plane tensors must agree (they do when
        // the caller indexes equally aligned tensors by the same row, as ReplayBuffer.store_many does).
        const uintptr_t sa = (uintptr_t)(state ? state : next_state), sb = (uintptr_t)(next_state ? next_state : state);
        i64 head = n;
        if (sa && (sa & 127u) == (sb & 127u)) head = (i64)((sa & 127u) / 8u);
        if (head > n) head = n;
        const i64 chunks = (n - head) / UC;
        ragged(0, head);
        if (chunks > 0) {
            const int grid = (int)(chunks < 4096 ? chunks : 4096);
            k_traj_unpack_c4_lines<<<dim3(grid), dim3(256), 0, st>>>(
                chunks, recs + head * 48, state ? state + head * 126 : nullptr, prob ? prob + head * 7 : nullptr,
                value ? value + head : nullptr, winner ? winner + head : nullptr,
                next_state ? next_state + head * 126 : nullptr, done ? done + head : nullptr);
        }
        ragged(head + chunks * UC, n - head - chunks * UC);
        AZ_HIP(hipGetLastError());
        return AZ_OK;
    }
    if (game == AZ_GAME_TTT) return unpack_lines_or_ragged<TTT>(n, recs, state, prob, value, winner, next_state, done, st);
    if (game == AZ_GAME_NBTTT) return unpack_lines_or_ragged<NBTTT>(n, recs, state, prob, value, winner, next_state, done, st);
    AZ_DISPATCH_GAME(game, {
        constexpr int PL = GT::GAME == AZ_GAME_C4 ? 126 : (GT::GAME == AZ_GAME_TTT ? 27 : 252);
        k_traj_unpack<GT><<<dim3(grid_for(n * PL)), dim3(256), 0, st>>>(n, recs, state, prob, value, winner, next_state,
                                                                       done);
    });
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

}  // namespace az
