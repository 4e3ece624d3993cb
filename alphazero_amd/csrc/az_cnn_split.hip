// az_cnn_split.hip -- K3 on the fp16 MFMA pipe with fp32-accurate products (AZ_K3_SPLIT; az_net_set_kernel).
//
// The same network (Connect4/Network.py:19-27,31-51,60-64,78-85; BatchNorm folded) as a DIRECT implicit GEMM on
// fp16 MFMA (v_mfma_f32_32x32x16_f16 in the tile loops), with every fp32 product a*w computed as hi*hi + hi*lo + lo*hi,
// hi/lo = the fp16 halves of the fp32 operands (22 mantissa bits; the dropped lo*lo term is 2^-22 relative), accumulated
// in fp32.
// RANGE (why no half can leave fp16's exponent range, whatever the weights and the position):
//   weights     folded in float64, scaled per layer by a power of two so that the largest is in [2^13, 2^14), split on the
//               host: hi <= 2^14 < 65504; a weight's lo half is exact down to 2^-24, i.e. 2^-38 of the layer's largest weight;
//   activations rescaled PER POSITION AND LAYER by a power of two 2^e, e chosen in the kernel from the largest magnitude
//               of that position's 42 x 128 values (one LDS ds_max_u32 per lane and column) so that it lands in
//               [2^13, 2^14); the consumer multiplies its fp32 accumulators by 2^-e (exact).  hi never overflows, and an
//               element's lo half is exact down to 2^-24 * 2^-e, i.e. 2^-37 of the position's largest activation -- far
//               below the 2^-24 relative rounding of the fp32 accumulation that every fp32 kernel has as well.
//   A position's scale depends on that position only, so its result does not depend on its neighbours in the batch.
// The activation image lives in LDS already split (hi plane + lo plane = the 4 bytes per element of an fp32
// image), channel-contiguous so that a B operand is one 16-byte LDS read.  3 positions per workgroup in the product form
// (126 cells on 128 MFMA columns); its 4 waves are a 2 x 2 grid of 64 output channels x 64 cells.  A lane keeps its own entries of x in fp32
// registers for the residual, t = SiLU(BN(conv3x3(x))) goes through the image and the 1x1 convolution is the same
// routine with one tap.  Prototypes and measurements: tools/ubench/{conv,block}_split_f16.hip, tools/k3_accuracy.py,
// tools/split_fp16_accuracy.py; DESIGN.md section 4.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "az_cnn.hpp"

namespace az {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int SCH = 128;                                 // channels
constexpr int SRS = SCH + 8;                             // halves per image row (cell): 128 channels + 8 of padding (272 B apart)
constexpr int SFRONT = 10;
// image rows for NS positions per workgroup: 7 rows x 9 cells per position (board rows 1..6, columns 1..7; a position's bottom
// zero row is the next one's top row) from row SFRONT on; cells run from row 20 to row 63 NS + 8 and a tap reaches 10 rows further
__host__ __device__ constexpr int srows(int ns) { return ns * 63 + 19; }
__host__ __device__ constexpr size_t split_lds(int ns) { return (size_t)2 * srows(ns) * SRS * 2; }
constexpr int SMFT = 4;                                  // 32-channel fragments in all
constexpr int SPLIT_DEFAULT_CFG = 34;   // measured in the bench (2 048 slots, 4 cohorts): 34 -> 822 games/s, 24 -> 696, 44 -> 677

struct CnnSDev {
    // tile weights, 32x32x16 fragments: [slice][kh][4 F][hi|lo][64 lanes] f16x8, lane = (channel 32 F + lane % 32,
    // k = channel 32 cs + 16 kh + 8 (lane / 32) + i)
    const f16x8* w_stem;   // 9 slices (tap), channels 0..2 real
    const f16x8* w3[3];    // 36 slices = tap * 4 + cs
    const f16x8* w1[3];    // 4 slices = cs
    const f16x8* w_head;   // [4 cs][hi|lo][64] : rows 0-2 policy conv, 3 value conv
    // the same tile weights as 16x16x32 fragments (k_cnn_forward_split16): [slice][8 F][hi|lo][64 lanes] f16x8, lane = (channel
    // 16 F + lane % 16, k = channel 32 cs + 8 (lane / 16) + i)
    const f16x8 *w16_stem, *w16_3[3], *w16_1[3];
    const float *b_stem, *b3[3], *b1[3], *b_head, *plin_w, *plin_b, *vlin_w, *vlin_b;
    float inv_stem, inv3[3], inv1[3], inv_head;
};
struct CnnS {
    CnnSDev d;
    std::vector<void*> allocs;
};

__host__ __device__ inline int scell(int s, int r, int c) { return SFRONT + s * 63 + (r + 1) * 9 + (c + 1); }
__device__ inline float ssilu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// sum over the 64 lanes, the same value in every lane: four DPP steps inside a row of 16 (quad swaps, then the half-row
// and row mirrors, which pair up groups whose lanes already agree), then the four rows through the scalar side
template <int CTRL>
__device__ inline float sdpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ inline float swave_sum(float v) {
    v = sdpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = sdpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = sdpp_add<0x141>(v);   // row_half_mirror
    v = sdpp_add<0x140>(v);   // row_mirror
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
#ifdef AZ_K3S_DBG_TS   // diagnostic builds only (tools/k3_variants.sh, tools/k3s_stamps.py): shader-clock stamps of one wave
__device__ unsigned long long g_split_ts[4][64];
#ifndef AZ_K3S_TS_BLOCK
#define AZ_K3S_TS_BLOCK 7
#endif
#define STAMP(i) do { if (blockIdx.x == AZ_K3S_TS_BLOCK && lane == 0) g_split_ts[wave & 3][i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#define SMFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)      // the tile loops
#define SMFMA16(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)    // the heads

// A convolution over the split image as 9 (3x3) or 1 (1x1) taps x PER_TAP k-slices of 32 channels; slice sl = tap * PER_TAP
// + cs.  The wave computes channel fragments fbase .. fbase + SMF - 1 for its 32 SNF cells.  The weight fragments of a slice
// come from L2 through the vector memory path, which serves the four waves of a CU one after the other: a wave tile of
// 64 channels x 64 (or 96) cells asks for half the weight bytes per MFMA that 128 x 48 does.
// NKH: k-steps of 16 per slice that hold anything (the stem's 3 input planes sit in the first one)
// The weights of a convolution's FIRST slice, requested by the caller before the barriers that publish the convolution's input
// image: they do not depend on it, and a convolution that starts by asking for them waits a whole L2 round trip with an idle
// matrix pipe (1-2 k cycles per convolution, seven convolutions).
template <int SMF>
struct WPre {
    f16x8 h[2][SMF], l[2][SMF];
};
template <int SMF, int NKH>
__device__ inline void w_prefetch(const f16x8* __restrict__ w, int nsl, int lane, int fbase, WPre<SMF>& p) {
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, nsl * 2 * SMFT * 2 * 1024, 0x00020000);
    const int wvo = lane * 16 + fbase * 2048;
#pragma unroll
    for (int kh = 0; kh < NKH; kh++)
#pragma unroll
        for (int f = 0; f < SMF; f++) {
            p.h[kh][f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, (kh * SMFT + f) * 2048, 0));
            p.l[kh][f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, (kh * SMFT + f) * 2048 + 1024, 0));
        }
}

// UNROLLED: every slice is its own copy of the code, so the tap offsets (LDS) and the slice offsets (weights) are immediates and
// the loop carries no address arithmetic at all; the weights come through buffer loads (SGPR descriptor + lane offset VGPR +
// scalar / immediate offset: no 64-bit vector adds).  LO_OFF: distance in halves from the hi plane to the lo plane.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int TAPS, int PER_TAP, bool LO_INPUT, int SMF, int SNF, int NST, int NKH = 2, bool UNROLLED = false>
__device__ inline void conv_split(const f16x8* __restrict__ w, const _Float16* hi, int lo_off, const int (&brow)[SNF],
                                  int lane, int fbase, f32x16 (&acc)[SMF][SNF], const WPre<SMF>& pre) {
    constexpr int NSL = TAPS * PER_TAP;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, NSL * 2 * SMFT * 2 * 1024, 0x00020000);
    const int wvo = lane * 16 + fbase * 2048;   // this lane's bytes inside a (slice, k-step) block of 4 fragments x hi | lo
    // v_mfma_f32_32x32x16_f16: a slice of 32 channels is two k-steps of 16 (kh); A = 32 channels x 16 k, B = 16 k x 32 cells
    // NST operand register sets: slice s + NST - 1 is requested while slice s is multiplied
    f16x8 ah[NST][2][SMF], al[NST][2][SMF], bh[NST][2][SNF], bl[NST][2][SNF];
    // operands of one slice into register set b: B from the image rows bb[j] (+ immediate k offset boff, in halves), A from the
    // weight block at byte offset wso (uniform) of the layer
    auto fetch_at = [&](const int (&bb)[SNF], int boff, int wso, int b, bool a_from_pre = false) {
#pragma unroll
        for (int kh = 0; kh < NKH; kh++) {
#pragma unroll
            for (int f = 0; f < SMF; f++) {
                if (a_from_pre) {   // slice 0: the caller asked for these before the barriers
                    ah[b][kh][f] = pre.h[kh][f];
                    al[b][kh][f] = pre.l[kh][f];
                    continue;
                }
#ifdef AZ_K3S_DBG_NOW   // diagnostic builds only (tools/k3_variants.sh): no weight traffic, results are garbage
                ah[b][kh][f] = (f16x8)((_Float16)(0.001f * (float)(wso + f)));
                al[b][kh][f] = (f16x8)((_Float16)(0.002f * (float)(wso + kh)));
#else
                const u32x4 qh = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, wso + (kh * SMFT + f) * 2048, 0);
                const u32x4 ql = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, wso + (kh * SMFT + f) * 2048 + 1024, 0);
                ah[b][kh][f] = __builtin_bit_cast(f16x8, qh);
                al[b][kh][f] = __builtin_bit_cast(f16x8, ql);
#endif
            }
#pragma unroll
            for (int j = 0; j < SNF; j++) {
#ifdef AZ_K3S_DBG_NOB   // diagnostic builds only: no LDS operand traffic, results are garbage
                bh[b][kh][j] = (f16x8)((_Float16)(0.25f + (float)((bb[j] + boff) & 3)));
                if (LO_INPUT) bl[b][kh][j] = (f16x8)((_Float16)(0.125f + (float)((bb[j] + boff) & 1)));
#else
                bh[b][kh][j] = *reinterpret_cast<const f16x8*>(hi + bb[j] + boff + 16 * kh);
                if (LO_INPUT) bl[b][kh][j] = *reinterpret_cast<const f16x8*>(hi + bb[j] + boff + 16 * kh + lo_off);
#endif
            }
        }
    };
    auto tap_off = [](int tap) { return ((tap / 3 - 1) * 9 + (tap % 3 - 1)) * SRS; };
    auto fetch = [&](int sl, int b, bool a_from_pre = false) {
        const int tap = TAPS == 1 ? 4 : sl / PER_TAP, cs = sl % PER_TAP;
        int bb[SNF];
#pragma unroll
        for (int j = 0; j < SNF; j++) bb[j] = brow[j] + tap_off(tap);
        fetch_at(bb, cs * 32, sl * (2 * SMFT * 2048), b, a_from_pre);
    };
    auto mfmas = [&](int b) {  // product by product: 6 independent accumulators between two uses of the same one
#ifdef AZ_K3S_DBG_LOADONLY   // diagnostic builds only: every operand is fetched as usual and kept alive, the MFMAs multiply constants
        // (results are garbage).  Against the NOW+NOB build (no fetches, constant operands) this prices the operand TRAFFIC alone;
        // against the product build it prices the bits toggling inside the matrix pipe.
        const f16x8 ca = (f16x8)((_Float16)0.25f), cb = (f16x8)((_Float16)0.5f);
#pragma unroll
        for (int kh = 0; kh < NKH; kh++) {
#pragma unroll
            for (int f = 0; f < SMF; f++) asm volatile("" ::"v"(ah[b][kh][f]), "v"(al[b][kh][f]));
#pragma unroll
            for (int j = 0; j < SNF; j++) {
                asm volatile("" ::"v"(bh[b][kh][j]));
                if (LO_INPUT) asm volatile("" ::"v"(bl[b][kh][j]));
            }
#pragma unroll
            for (int rep = 0; rep < (LO_INPUT ? 3 : 2); rep++)
#pragma unroll
                for (int f = 0; f < SMF; f++)
#pragma unroll
                    for (int j = 0; j < SNF; j++) SMFMA(ca, cb, acc[f][j]);
        }
#else
#pragma unroll
        for (int kh = 0; kh < NKH; kh++) {
#pragma unroll
            for (int f = 0; f < SMF; f++)
#pragma unroll
                for (int j = 0; j < SNF; j++) SMFMA(ah[b][kh][f], bh[b][kh][j], acc[f][j]);
            if (LO_INPUT) {
#pragma unroll
                for (int f = 0; f < SMF; f++)
#pragma unroll
                    for (int j = 0; j < SNF; j++) SMFMA(ah[b][kh][f], bl[b][kh][j], acc[f][j]);
            }
#pragma unroll
            for (int f = 0; f < SMF; f++)
#pragma unroll
                for (int j = 0; j < SNF; j++) SMFMA(al[b][kh][f], bh[b][kh][j], acc[f][j]);
        }
#endif
    };
    // One basic block per pair of slices: the fetches are UNCONDITIONAL (the slice index is clamped; the last pair of an even
    // count fetches the last slice once more), because a conditional fetch is a block of 20 loads of its own during which the
    // matrix pipe idles, and the scheduler cannot move loads across the branch.  Inside the block the loads of the next slice
    // are dealt out one per MFMA (sched_group_barrier pipeline: MFMA, VMEM read | MFMA, DS read | ... | the remaining MFMAs),
    // so a load instruction issues under an MFMA instead of in a burst between two slices.
    constexpr int NA = 2 * NKH * SMF, NB = (LO_INPUT ? 2 : 1) * NKH * SNF, NM = NKH * (LO_INPUT ? 3 : 2) * SMF * SNF;
    static_assert(NM >= NA + NB, "fewer MFMAs than loads per slice");
    auto deal = [&]() {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - NA - NB, 0);
    };
    static_assert(NSL >= NST, "fewer slices than operand register sets");
#pragma unroll
    for (int b = 0; b + 1 < NST; b++) fetch(b, b, b == 0);
    if constexpr (UNROLLED && TAPS == 9 && PER_TAP == 4) {
        // one trip per tap, its four slices unrolled (two register sets): the image rows of this tap and of the next one are
        // kept as per-lane bases, so that a slice's operands are base + immediate; 2 SNF vector adds per tap, none per slice
        static_assert(NST == 2, "the tap-unrolled form rotates two register sets");
        int bcur[SNF], bnxt[SNF];
#pragma unroll
        for (int j = 0; j < SNF; j++) bcur[j] = brow[j] + tap_off(0), bnxt[j] = brow[j] + tap_off(1);
        int wso = 0;
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
#pragma unroll
            for (int cs = 0; cs < 4; cs++) {
                if (cs < 3) fetch_at(bcur, (cs + 1) * 32, wso + (cs + 1) * (2 * SMFT * 2048), (cs + 1) & 1);
                else fetch_at(bnxt, 0, tap < 8 ? wso + 4 * (2 * SMFT * 2048) : wso, 0);   // after the last tap: a valid slice again, unused
                mfmas(cs & 1);
                deal();
            }
            wso += 4 * (2 * SMFT * 2048);
            const int t2 = tap + 2 < 9 ? tap + 2 : 8;
#pragma unroll
            for (int j = 0; j < SNF; j++) bcur[j] = bnxt[j], bnxt[j] = brow[j] + tap_off(t2);
        }
    } else if constexpr (UNROLLED) {
#pragma unroll
        for (int sl = 0; sl < NSL; sl++) {
            if (sl + NST - 1 < NSL) fetch(sl + NST - 1, (sl + NST - 1) % NST);
            mfmas(sl % NST);
            if (sl + NST - 1 < NSL) deal();
        }
    } else {
#pragma unroll 1
        for (int t = 0; t < NSL / NST; t++) {
            const int sl = NST * t;
#pragma unroll
            for (int b = 0; b < NST; b++) {
                const int nx = sl + b + NST - 1;
                fetch(nx < NSL ? nx : NSL - 1, (b + NST - 1) % NST);
                mfmas(b);
                deal();
            }
        }
        // the NSL % NST last slices sit in register sets 0 .. (their fetches were issued by the last trip of the loop)
#pragma unroll
        for (int b = 0; b < NSL % NST; b++) mfmas(b);
    }
}

// power-of-two scale that brings a position's largest magnitude m into [2^13, 2^14), and its inverse (both exact).
// m's biased exponent E is clamped to [20, 240] (m outside 2^-107 .. 2^113 is no network any more; m == 0: all zero).
__device__ inline void range_scale(float m, float& sc, float& isc) {
    int E = (int)((__float_as_uint(m) >> 23) & 0xFFu);
    E = E < 20 ? 20 : (E > 240 ? 240 : E);
    sc = __uint_as_float((u32)(267 - E) << 23);    // 2^(13 - (E - 127))
    isc = __uint_as_float((u32)(E - 13) << 23);    // 2^((E - 127) - 13)
}

// this lane's values, times the position's scale -> split -> image, in place.  32x32 accumulator: register 4 q + r of
// fragment f is channel ch0 + 32 f + 8 q + r (ch0 = 64 wa + 4 (lane / 32)) of cell column lane % 32
template <int SMF, int SNF>
__device__ inline void store_image(_Float16* hi, _Float16* lo, const int (&crow)[SNF], const bool (&live)[SNF], int ch0,
                                   const f32x16 (&val)[SMF][SNF], const float (&sc)[SNF]) {
#pragma unroll
    for (int f = 0; f < SMF; f++)
#pragma unroll
        for (int j = 0; j < SNF; j++)
            if (live[j]) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f16x4 oh, ol;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float v = val[f][j][4 * q + r] * sc[j];
                        oh[r] = (_Float16)v;
                        ol[r] = (_Float16)(v - (float)oh[r]);
                    }
                    const int at = crow[j] + ch0 + 32 * f + 8 * q;
                    *reinterpret_cast<f16x4*>(hi + at) = oh;
                    *reinterpret_cast<f16x4*>(lo + at) = ol;
                }
            }
}

// NS positions per workgroup, NW waves.  The 42 NS cells are cut into NWB column groups of 32 SNF, the 128 channels into
// NW / NWB groups; a wave owns one (channel group, column group) tile.  <3, 4> is the product form (SPLIT_DEFAULT_CFG); the
// others are kept for measurement (AZ_K3_SPLIT_CFG; DESIGN.md 4.2 has all four side by side).
//   <4, 4>  one workgroup per CU, one wave per SIMD: wave tile 64 channels x 96 cells, up to 512 registers per lane
//   <2, 4>  TWO workgroups per CU (80 KB of LDS and 256 registers per lane each): wave tile 32 channels x 96 cells (84 live).  A
//           SIMD holds one wave of each workgroup, so one workgroup's epilogue arithmetic (SiLU, split, image stores) and operand
//           waits run under the other's MFMAs -- a wave's own VALU work does not overlap its MFMAs on this chip
//   <3, 4>  one workgroup per CU, 126 cells on 128 columns (1.6 % of the MFMA work on padding columns instead of 12.5 %): wave tile
//           64 channels x 64 cells; a third more weight bytes per position
//   <4, 8>  one workgroup of 8 waves (32 x 96 tiles): kept for measurement; all waves share the barriers, so nothing overlaps
// the workgroup's whole pass over the NS positions that start at batch index s0 (a `return` ends the pass of the wave that takes it)
template <int NS, int NW>
__device__ __forceinline__ void split_pass(const CnnSDev& w, i64 n_host, const u32* __restrict__ n_ptr,
                                           const u32* __restrict__ list, const u64* __restrict__ states,
                                           float* __restrict__ logp, float* __restrict__ logv,
                                           float* __restrict__ probs, float* __restrict__ value, int desync,
                                           u32 max_n, const i64 s0) {
    extern __shared__ _Float16 lds[];
    constexpr int SNS = NS, SCELLS = NS * 42, SROWS = srows(NS);
    __shared__ float hb[4][SCELLS];  // head activations: 3 policy channels, 1 value channel
    __shared__ float s_player[SNS];
    __shared__ long long s_slot[SNS];
    __shared__ u32 s_amax[7][SNS];   // per layer output and position: bits of the largest |activation| (ds_max_u32)
    _Float16* hi = lds;
    _Float16* lo = lds + SROWS * SRS;
    // waves along cells / channels; 32-cell and 32-channel fragments per wave
    constexpr int NWB = NS == 2 ? 1 : 2, SNF = NS == 3 ? 2 : 3, NWA = NW / NWB, SMF = 4 / NWA;
    constexpr int NST = 2;                 // operand register sets (a third, tried on the 64 x 64 tile, measured within noise)
    constexpr bool UNR = NS == 3;          // slice loops unrolled (the product configuration)
    static_assert(NWA * NWB == NW && NWA * SMF == 4 && NWB * SNF * 32 >= SCELLS, "wave grid must tile 128 channels x the cells");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, g = lane >> 4;
    STAMP(30);   // kernel entry
    // ---- set-up.  Three dependent global reads stand between a workgroup and its first MFMA (leaf count -> leaf list -> position);
    // the count and the list entry are requested together (the list holds max_n entries whatever the count says), the image is
    // cleared while they travel, and the position's words are requested before the second half of the clearing.
    WPre<SMF> pre;   // first-slice weights of the next convolution, requested ahead of the barriers (w_prefetch)
    w_prefetch<SMF, 1>(w.w_stem, 9, lane, SMF * (wave % NWA), pre);   // the stem's, under the image set-up
    const int ps = tid / 42, pq = tid % 42;                   // thread -> (position, cell) of the input planes, tid < SCELLS
    const bool mine = tid < SCELLS;
    u32 li = 0u;
    if (list && mine && s0 + ps < (i64)max_n) li = list[s0 + ps];
    const i64 n = n_ptr ? (i64)(*n_ptr) : n_host;
    if (NS == 2 && desync > 0) {
        // Two workgroups share a CU and one wave of each shares every SIMD.  Started together they would walk the layers in
        // step and their epilogues would coincide; the workgroup whose waves sit in the odd wave slots starts `desync` x 8 k
        // cycles late, so that its arithmetic phases fall under the other's MFMA phases.
        const u32 slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 1u;   // HW_ID.WAVE_ID bit 0
        if (slot)
            for (int i = 0; i < desync; i++) __builtin_amdgcn_s_sleep(127);
    }
    constexpr int NZ = SROWS * SRS / 4;    // the image (both planes) in 16-byte units
    static_assert(SROWS * SRS % 4 == 0, "image size in 16-byte units");
    u32x4* const lds16 = reinterpret_cast<u32x4*>(lds);
    for (int i = tid; i < NZ / 2; i += NW * 64) lds16[i] = (u32x4){0u, 0u, 0u, 0u};
    if (s0 >= n) return;
    const bool have = mine && s0 + ps < n;
    const i64 myslot = list ? (i64)li : (s0 + ps);
    u64 w0 = 0, w1 = 0;
    if (have) {
        w0 = states[myslot * 2];
        w1 = states[myslot * 2 + 1];
    }
    for (int i = NZ / 2 + tid; i < NZ; i += NW * 64) lds16[i] = (u32x4){0u, 0u, 0u, 0u};
    if (tid < 7 * SNS) (&s_amax[0][0])[tid] = 0u;
    __syncthreads();
    // ---- input planes from bitboards (env.current_state, env_cython.pyx:143-149): channels 0..2, exact in fp16
    if (have) {
        const int row = pq / 7, col = pq % 7;
        const int bit = col * 7 + (5 - row);
        const float vt = (w1 >> 63) ? -1.f : 1.f;
        _Float16* p = hi + scell(ps, row, col) * SRS;
        p[0] = (_Float16)(float)((w0 >> bit) & 1ull);
        p[1] = (_Float16)(float)((w1 >> bit) & 1ull);
        p[2] = (_Float16)vt;
        if (pq == 0) {
            s_player[ps] = vt;
            s_slot[ps] = myslot;
        }
    }
    __syncthreads();
    // wave (wa, wb): channels 32 SMF wa .. (32-channel fragments SMF wa ..) x cells 32 SNF wb ..
    const int wa = wave % NWA, wb = wave / NWA, fbase = SMF * wa, l32 = lane & 31, kg = lane >> 5, ch0 = 32 * SMF * wa + 4 * kg;
    int brow[SNF], crow[SNF], cpos[SNF];
    bool live[SNF];
    float sc[SNF], isc[SNF];   // scale of the image's current contents for this lane's columns (its position's), and 1/scale
#pragma unroll
    for (int j = 0; j < SNF; j++) {
        int col = wb * (32 * SNF) + j * 32 + l32;
        live[j] = col < SCELLS;
        if (!live[j]) col = 0;
        const int s = col / 42, p = col % 42;
        cpos[j] = s;
        crow[j] = scell(s, p / 7, p % 7) * SRS;
        brow[j] = crow[j] + 8 * kg;
        sc[j] = 1.f;    // the input planes are 0 / +-1
        isc[j] = 1.f;
    }
    f32x16 acc[SMF][SNF];
    auto zero = [&](f32x16 (&a)[SMF][SNF]) {
#pragma unroll
        for (int f = 0; f < SMF; f++)
#pragma unroll
            for (int j = 0; j < SNF; j++)
#pragma unroll
                for (int i = 0; i < 16; i++) a[f][j][i] = 0.f;
    };
    // this lane's 16 SMF folded BatchNorm biases of a layer, requested BEFORE the layer's convolution so that the epilogue does
    // not open with a round trip to L2
    f32x4 bia[SMF][4];
    auto load_bias = [&](const float* bias) {
#pragma unroll
        for (int f = 0; f < SMF; f++)
#pragma unroll
            for (int q = 0; q < 4; q++) bia[f][q] = *reinterpret_cast<const f32x4*>(bias + ch0 + 32 * f + 8 * q);
    };
    // a = SiLU(a * (inv / scale of the image the convolution read) + bias[channel] (+ res))
    auto activate = [&](f32x16 (&a)[SMF][SNF], float inv, const f32x16 (*res)[SNF]) {
        // two values at a time on the packed fp32 instructions (v_pk_fma / v_pk_mul / v_pk_add: two values for the issue slot
        // of one); per pair: fma, [add residual], mul by -log2 e, 2 x v_exp, add 1, 2 x v_rcp, mul
        f32x2 invj[SNF];
#pragma unroll
        for (int j = 0; j < SNF; j++) invj[j] = (f32x2){inv * isc[j], inv * isc[j]};
        const f32x2 nl2e = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.f, 1.f};
#pragma unroll
        for (int f = 0; f < SMF; f++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 bb = bia[f][q];
#pragma unroll
                for (int j = 0; j < SNF; j++)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 a2 = {a[f][j][4 * q + r], a[f][j][4 * q + r + 1]};
                        f32x2 v = a2 * invj[j] + (f32x2){bb[r], bb[r + 1]};
                        if (res) v += (f32x2){res[f][j][4 * q + r], res[f][j][4 * q + r + 1]};
                        const f32x2 t = v * nl2e;
                        const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + one;
                        const f32x2 o = v * (f32x2){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                        a[f][j][4 * q + r] = o[0];
                        a[f][j][4 * q + r + 1] = o[1];
                    }
            }
    };
    // the layer's output a -> image: every position's largest magnitude through LDS, scale, split, store.  The first
    // barrier is also the one after which no wave reads the image's old contents any more.
    auto to_image = [&](int layer, const f32x16 (&a)[SMF][SNF]) {
#pragma unroll
        for (int j = 0; j < SNF; j++) {
            float m = 0.f;
#pragma unroll
            for (int f = 0; f < SMF; f++)
#pragma unroll
                for (int i = 0; i < 16; i++) m = fmaxf(m, fabsf(a[f][j][i]));
            if (live[j]) atomicMax(&s_amax[layer][cpos[j]], __float_as_uint(m));
        }
        if (layer == 3) STAMP(32);
        __syncthreads();
        if (layer == 3) STAMP(33);
#pragma unroll
        for (int j = 0; j < SNF; j++) range_scale(__uint_as_float(s_amax[layer][cpos[j]]), sc[j], isc[j]);
        store_image<SMF, SNF>(hi, lo, crow, live, ch0, a, sc);
        if (layer == 3) STAMP(34);
        __syncthreads();
    };
    // ---- stem: conv3x3(3->128) + BN + SiLU (Network.py:33-36); the input has no lo half
    STAMP(0);
    zero(acc);
    load_bias(w.b_stem);
    conv_split<9, 1, false, SMF, SNF, NST, 1, false>(w.w_stem, hi, SROWS * SRS, brow, lane, fbase, acc, pre);
    STAMP(1);
    activate(acc, w.inv_stem, nullptr);
    STAMP(2);
    // this lane's own entries of x in fp32, kept for the residual while the image holds t (what the image holds is x rounded
    // to the 22 bits of a hi / lo pair; the residual takes the unrounded value)
    f32x16 xres[SMF][SNF];
    auto keep = [&](const f32x16 (&a)[SMF][SNF]) {
#pragma unroll
        for (int f = 0; f < SMF; f++)
#pragma unroll
            for (int j = 0; j < SNF; j++) xres[f][j] = a[f][j];
    };
    keep(acc);
    w_prefetch<SMF, 2>(w.w3[0], 36, lane, fbase, pre);
    to_image(0, acc);
    STAMP(3);

    // ---- 3 residual blocks (Network.py:19-27): x = SiLU(x + BN(conv1x1(SiLU(BN(conv3x3(x))))))
    for (int blk = 0; blk < 3; blk++) {
        STAMP(4 + 8 * blk);
        zero(acc);
        load_bias(w.b3[blk]);
        conv_split<9, 4, true, SMF, SNF, NST, 2, UNR>(w.w3[blk], hi, SROWS * SRS, brow, lane, fbase, acc, pre);
        STAMP(5 + 8 * blk);       // 3x3 convolution
        activate(acc, w.inv3[blk], nullptr);
        STAMP(6 + 8 * blk);       // bias + SiLU
        w_prefetch<SMF, 2>(w.w1[blk], 4, lane, fbase, pre);
        to_image(1 + 2 * blk, acc);  // t
        STAMP(7 + 8 * blk);       // range, split, store, two barriers
        zero(acc);
        load_bias(w.b1[blk]);
        conv_split<1, 4, true, SMF, SNF, NST, 2, UNR>(w.w1[blk], hi, SROWS * SRS, brow, lane, fbase, acc, pre);
        STAMP(8 + 8 * blk);       // 1x1 convolution
        activate(acc, w.inv1[blk], xres);
        STAMP(9 + 8 * blk);
        keep(acc);
        if (blk < 2) {
            w_prefetch<SMF, 2>(w.w3[blk + 1], 36, lane, fbase, pre);
        } else if constexpr (SMF == 2) {
            // the heads' convolution weights (8 fragments per lane) travel in the registers of the first-slice weights
#pragma unroll
            for (int cs = 0; cs < 4; cs++) {
                pre.h[cs >> 1][cs & 1] = w.w_head[(size_t)(cs * 2 + 0) * 64 + lane];
                pre.l[cs >> 1][cs & 1] = w.w_head[(size_t)(cs * 2 + 1) * 64 + lane];
            }
        }
        to_image(2 + 2 * blk, acc);
        STAMP(10 + 8 * blk);
    }
    // what the end of the network reads from memory, requested before the heads' MFMAs: the heads' biases, and for the waves that
    // finish a position (wave s, position s) that position's share of the Linear weights -- lane j holds the weights of the
    // flattened inputs j and j + 64 (nn.Flatten of NCHW: c*42 + h*7 + w)
    float pw[2][7], vw[3], hbias[4], pbias[7], vbias[3];
#pragma unroll
    for (int r = 0; r < 4; r++) hbias[r] = w.b_head[r];
#pragma unroll
    for (int a = 0; a < 7; a++) pbias[a] = w.plin_b[a];
#pragma unroll
    for (int k = 0; k < 3; k++) vbias[k] = w.vlin_b[k];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int a = 0; a < 7; a++) pw[i][a] = (wave < SNS && lane + 64 * i < 126) ? w.plin_w[a * 126 + lane + 64 * i] : 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) vw[k] = (wave < SNS && lane < 42) ? w.vlin_w[k * 42 + lane] : 0.f;

    STAMP(28);
    // ---- heads: conv1x1(128->3) / conv1x1(128->1) + BN + SiLU (Network.py:38-51) as one 16-row fragment of
    // v_mfma_f32_16x16x32_f16 (cell = lane % 16, k group = lane / 16); the 2 SNF groups of 16 cells of a wave column are
    // dealt to the NWA waves of that column (all to the first one where they do not divide)
    constexpr bool HSHARE = (2 * SNF) % NWA == 0;
    constexpr int HJ = HSHARE ? 2 * SNF / NWA : 2 * SNF;
    if (HSHARE || wa == 0) {
#pragma unroll
        for (int j0 = 0; j0 < HJ; j0++) {
            const int jj = HSHARE ? wa * HJ + j0 : j0;
            int col = wb * (32 * SNF) + jj * 16 + n16;
            const bool lv = col < SCELLS;
            if (!lv) col = 0;
            const int sp = col / 42, p = col % 42;
            const int b16 = scell(sp, p / 7, p % 7) * SRS + 8 * g;
            f32x4 a4 = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int cs = 0; cs < 4; cs++) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(hi + b16 + cs * 32);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(lo + b16 + cs * 32);
                f16x8 wh, wl;
                if constexpr (SMF == 2) {
                    wh = pre.h[cs >> 1][cs & 1];
                    wl = pre.l[cs >> 1][cs & 1];
                } else {
                    wh = w.w_head[(size_t)(cs * 2 + 0) * 64 + lane];
                    wl = w.w_head[(size_t)(cs * 2 + 1) * 64 + lane];
                }
                SMFMA16(wh, bh, a4);
                SMFMA16(wh, bl, a4);
                SMFMA16(wl, bh, a4);
            }
            if (g == 0 && lv) {
                float hsc, hisc;   // the image holds the last block's output at its position's scale
                range_scale(__uint_as_float(s_amax[6][sp]), hsc, hisc);
#pragma unroll
                for (int r = 0; r < 4; r++) hb[r][col] = ssilu(a4[r] * (w.inv_head * hisc) + hbias[r]);
            }
        }
    }
    __syncthreads();
    STAMP(29);
    // ---- Linear(126->7) / Linear(42->3) + log-softmax + predict() (Network.py:60-64,78-85): wave s, position s
    const int s = wave;
    if (s >= SNS || s0 + s >= n) return;
    float pl[7], vl[3];
    {
        const float x0 = hb[lane / 42][s * 42 + lane % 42];                                   // flattened input j = lane
        const float x1 = lane + 64 < 126 ? hb[(lane + 64) / 42][s * 42 + (lane + 64) % 42] : 0.f;
        const float xv = lane < 42 ? hb[3][s * 42 + lane] : 0.f;
#pragma unroll
        for (int a = 0; a < 7; a++) pl[a] = swave_sum(x0 * pw[0][a] + x1 * pw[1][a]) + pbias[a];
#pragma unroll
        for (int k = 0; k < 3; k++) vl[k] = swave_sum(xv * vw[k]) + vbias[k];
    }
    // every lane holds the 7 + 3 sums; the exponentials inside the log-sum-exps are the hardware's (their rounding enters the
    // log-probabilities at 1e-7), the logarithms are libm's; lane a < 7 finishes action a, lanes 8..10 the value classes
    float mx = pl[0];
#pragma unroll
    for (int a = 1; a < 7; a++) mx = fmaxf(mx, pl[a]);
    float se = 0.f;
#pragma unroll
    for (int a = 0; a < 7; a++) se += __expf(pl[a] - mx);
    const float lse = logf(se);
    const float mv = fmaxf(vl[0], fmaxf(vl[1], vl[2]));
    const float lsv = logf(__expf(vl[0] - mv) + __expf(vl[1] - mv) + __expf(vl[2] - mv));
    const i64 slot = s_slot[s];
    if (lane < 7) {
        float mine_l = pl[0];
#pragma unroll
        for (int a = 1; a < 7; a++) mine_l = lane == a ? pl[a] : mine_l;
        const float lp = mine_l - mx - lse;
        if (logp) logp[slot * 7 + lane] = lp;
        if (probs) probs[slot * 7 + lane] = __expf(lp);
    } else if (lane >= 8 && lane < 11) {
        const int k = lane - 8;
        const float lv = (k == 0 ? vl[0] : (k == 1 ? vl[1] : vl[2])) - mv - lsv;
        if (logv) logv[slot * 3 + k] = lv;
    } else if (lane == 16) {
        // Network.py:83-84: classes [draw, mover wins, mover loses]
        const float v0 = __expf(vl[0] - mv - lsv), v1 = __expf(vl[1] - mv - lsv), v2 = __expf(vl[2] - mv - lsv);
        if (value) value[slot] = (-s_player[s]) * 0.5f * v0 + v1 - v2;
    }
    STAMP(31);   // this wave's end
}
template <int NS, int NW>
__global__ __launch_bounds__(NW * 64, NS == 2 ? 2 : 1) void k_cnn_forward_split(CnnSDev w, i64 n_host, const u32* __restrict__ n_ptr,
                                                              const u32* __restrict__ list, const u64* __restrict__ states,
                                                              float* __restrict__ logp, float* __restrict__ logv,
                                                              float* __restrict__ probs, float* __restrict__ value, int desync,
                                                              u32 max_n) {
    split_pass<NS, NW>(w, n_host, n_ptr, list, states, logp, logv, probs, value, desync, max_n, (i64)blockIdx.x * NS);
}
// The same with a grid SMALLER than the batch may need: a workgroup takes every gridDim.x-th group of NS positions.  For launches
// whose capacity is far above what they usually hold (speculative rows: 33 rows per game, a handful filled): a workgroup that
// finds nothing to do still costs the dispatcher ~0.1 us, in sequence, and thousands of them cost more than the launch's work
// (rolling self-play at 256 slots: 2 816 workgroups for ~80 rows, 140 us per launch instead of 70).
template <int NS, int NW>
__global__ __launch_bounds__(NW * 64, NS == 2 ? 2 : 1) void k_cnn_forward_split_strided(CnnSDev w, const u32* __restrict__ n_ptr,
                                                              const u32* __restrict__ list, const u64* __restrict__ states,
                                                              float* __restrict__ probs, float* __restrict__ value, u32 max_n) {
    const i64 n = (i64)(*n_ptr);
    for (i64 s0 = (i64)blockIdx.x * NS; s0 < n; s0 += (i64)gridDim.x * NS) {
        split_pass<NS, NW>(w, 0, n_ptr, list, states, nullptr, nullptr, probs, value, 0, max_n, s0);
        __syncthreads();   // the pass's last readers of the head activations and slot numbers, before the next pass's writers
    }
}

// ------------------------------------------------------------------------------------------------ 16x16x32 form
// The product configuration (3 positions per workgroup, 4 waves as a 2 x 2 grid of 64 channels x 64 cells) with the tile loops
// on v_mfma_f32_16x16x32_f16: 4 x 4 fragments of 16 x 16 per wave (the same 64 accumulator registers), one k-step of 32 per
// 32-channel slice, the same 8 + 8 operand loads per slice.  Built for the A/B that VERDICT round 2 (item 3) asks for
// (MI355X_MICROARCH.md "DVFS give-back" (7): under the power limit the clock can depend on the MFMA shape); selected with
// AZ_K3_SPLIT_CFG=316.  Everything outside the tile loops -- image layout, range scaling, heads, outputs -- is the 32x32x16
// kernel's, re-indexed for the 16 x 16 accumulator (4 consecutive channels of one cell per fragment and lane).
struct WPre16 {
    f16x8 h[4], l[4];
};
__device__ inline void w_prefetch16(const f16x8* __restrict__ w, int nsl, int lane, int fbase, WPre16& p) {
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, nsl * 16384, 0x00020000);
    const int wvo = lane * 16 + fbase * 2048;
#pragma unroll
    for (int f = 0; f < 4; f++) {
        p.h[f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, f * 2048, 0));
        p.l[f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, f * 2048 + 1024, 0));
    }
}

template <int TAPS, int PER_TAP, bool LO_INPUT>
__device__ inline void conv_split16(const f16x8* __restrict__ w, const _Float16* hi, int lo_off, const int (&brow)[4], int lane,
                                    int fbase, f32x4 (&acc)[4][4], const WPre16& pre) {
    constexpr int NSL = TAPS * PER_TAP;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, NSL * 16384, 0x00020000);
    const int wvo = lane * 16 + fbase * 2048;
    f16x8 ah[2][4], al[2][4], bh[2][4], bl[2][4];
    auto fetch_at = [&](const int (&bb)[4], int boff, int wso, int b, bool a_from_pre = false) {
#pragma unroll
        for (int f = 0; f < 4; f++) {
            if (a_from_pre) {
                ah[b][f] = pre.h[f];
                al[b][f] = pre.l[f];
                continue;
            }
            ah[b][f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, wso + f * 2048, 0));
            al[b][f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, wso + f * 2048 + 1024, 0));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            bh[b][j] = *reinterpret_cast<const f16x8*>(hi + bb[j] + boff);
            if (LO_INPUT) bl[b][j] = *reinterpret_cast<const f16x8*>(hi + bb[j] + boff + lo_off);
        }
    };
    auto tap_off = [](int tap) { return ((tap / 3 - 1) * 9 + (tap % 3 - 1)) * SRS; };
    // cell fragment outer, channel fragment inner: consecutive MFMAs share their B operand (tools/ubench/mfma_shape.hip: 19.5
    // cycles per MFMA in that order against 22 with the A operand shared)
    auto mfmas = [&](int b) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int f = 0; f < 4; f++) SMFMA16(ah[b][f], bh[b][j], acc[f][j]);
        if (LO_INPUT) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int f = 0; f < 4; f++) SMFMA16(ah[b][f], bl[b][j], acc[f][j]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int f = 0; f < 4; f++) SMFMA16(al[b][f], bh[b][j], acc[f][j]);
    };
    constexpr int NA = 8, NB = LO_INPUT ? 8 : 4, NM = (LO_INPUT ? 3 : 2) * 16;
    auto deal = [&]() {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - NA - NB, 0);
    };
    if constexpr (TAPS == 9 && PER_TAP == 4) {
        int bcur[4], bnxt[4];
#pragma unroll
        for (int j = 0; j < 4; j++) bcur[j] = brow[j] + tap_off(0), bnxt[j] = brow[j] + tap_off(1);
        fetch_at(bcur, 0, 0, 0, true);
        int wso = 0;
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
#pragma unroll
            for (int cs = 0; cs < 4; cs++) {
                if (cs < 3) fetch_at(bcur, (cs + 1) * 32, wso + (cs + 1) * 16384, (cs + 1) & 1);
                else fetch_at(bnxt, 0, tap < 8 ? wso + 4 * 16384 : wso, 0);
                mfmas(cs & 1);
                deal();
            }
            wso += 4 * 16384;
            const int t2 = tap + 2 < 9 ? tap + 2 : 8;
#pragma unroll
            for (int j = 0; j < 4; j++) bcur[j] = bnxt[j], bnxt[j] = brow[j] + tap_off(t2);
        }
    } else {
        auto fetch = [&](int sl, int b, bool a_from_pre = false) {
            const int tap = TAPS == 1 ? 4 : sl / PER_TAP, cs = sl % PER_TAP;
            int bb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) bb[j] = brow[j] + tap_off(tap);
            fetch_at(bb, cs * 32, sl * 16384, b, a_from_pre);
        };
        fetch(0, 0, true);
#pragma unroll
        for (int sl = 0; sl < NSL; sl++) {
            if (sl + 1 < NSL) fetch(sl + 1, (sl + 1) & 1);
            mfmas(sl & 1);
            if (sl + 1 < NSL) deal();
        }
    }
}

__global__ __launch_bounds__(256, 1) void k_cnn_forward_split16(CnnSDev w, i64 n_host, const u32* __restrict__ n_ptr,
                                                               const u32* __restrict__ list, const u64* __restrict__ states,
                                                               float* __restrict__ logp, float* __restrict__ logv,
                                                               float* __restrict__ probs, float* __restrict__ value, u32 max_n) {
    extern __shared__ _Float16 lds[];
    constexpr int SNS = 3, SCELLS = 126, SROWS = srows(3);
    __shared__ float hb[4][SCELLS];
    __shared__ float s_player[SNS];
    __shared__ long long s_slot[SNS];
    __shared__ u32 s_amax[7][SNS];
    _Float16* hi = lds;
    _Float16* lo = lds + SROWS * SRS;
    const i64 s0 = (i64)blockIdx.x * SNS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, g = lane >> 4;
    const int wa = wave & 1, wb = wave >> 1, fbase = 4 * wa, ch0 = 64 * wa + 4 * g;
    WPre16 pre;
    w_prefetch16(w.w16_stem, 9, lane, fbase, pre);
    const int ps = tid / 42, pq = tid % 42;
    const bool mine = tid < SCELLS;
    u32 li = 0u;
    if (list && mine && s0 + ps < (i64)max_n) li = list[s0 + ps];
    const i64 n = n_ptr ? (i64)(*n_ptr) : n_host;
    constexpr int NZ = SROWS * SRS / 4;
    u32x4* const lds16 = reinterpret_cast<u32x4*>(lds);
    for (int i = tid; i < NZ / 2; i += 256) lds16[i] = (u32x4){0u, 0u, 0u, 0u};
    if (s0 >= n) return;
    const bool have = mine && s0 + ps < n;
    const i64 myslot = list ? (i64)li : (s0 + ps);
    u64 w0 = 0, w1 = 0;
    if (have) {
        w0 = states[myslot * 2];
        w1 = states[myslot * 2 + 1];
    }
    for (int i = NZ / 2 + tid; i < NZ; i += 256) lds16[i] = (u32x4){0u, 0u, 0u, 0u};
    if (tid < 7 * SNS) (&s_amax[0][0])[tid] = 0u;
    __syncthreads();
    if (have) {   // input planes from bitboards (env.current_state, env_cython.pyx:143-149)
        const int row = pq / 7, col = pq % 7;
        const int bit = col * 7 + (5 - row);
        const float vt = (w1 >> 63) ? -1.f : 1.f;
        _Float16* p = hi + scell(ps, row, col) * SRS;
        p[0] = (_Float16)(float)((w0 >> bit) & 1ull);
        p[1] = (_Float16)(float)((w1 >> bit) & 1ull);
        p[2] = (_Float16)vt;
        if (pq == 0) {
            s_player[ps] = vt;
            s_slot[ps] = myslot;
        }
    }
    __syncthreads();
    // fragment (f, j) of wave (wa, wb): channels 64 wa + 16 f .. + 15, cells 64 wb + 16 j .. + 15; accumulator register r of
    // lane (n16, g) = channel 64 wa + 16 f + 4 g + r of cell 64 wb + 16 j + n16
    int brow[4], crow[4], cpos[4];
    bool live[4];
    float sc[4], isc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int col = wb * 64 + j * 16 + n16;
        live[j] = col < SCELLS;
        if (!live[j]) col = 0;
        const int s = col / 42, p = col % 42;
        cpos[j] = s;
        crow[j] = scell(s, p / 7, p % 7) * SRS;
        brow[j] = crow[j] + 8 * g;
        sc[j] = 1.f;
        isc[j] = 1.f;
    }
    f32x4 acc[4][4];
    auto zero = [&]() {
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    f32x4 bia[4];
    auto load_bias = [&](const float* bias) {
#pragma unroll
        for (int f = 0; f < 4; f++) bia[f] = *reinterpret_cast<const f32x4*>(bias + ch0 + 16 * f);
    };
    f32x4 xres[4][4];
    auto activate = [&](float inv, bool with_res) {
        f32x2 invj[4];
#pragma unroll
        for (int j = 0; j < 4; j++) invj[j] = (f32x2){inv * isc[j], inv * isc[j]};
        const f32x2 nl2e = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.f, 1.f};
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 a2 = {acc[f][j][r], acc[f][j][r + 1]};
                    f32x2 v = a2 * invj[j] + (f32x2){bia[f][r], bia[f][r + 1]};
                    if (with_res) v += (f32x2){xres[f][j][r], xres[f][j][r + 1]};
                    const f32x2 t = v * nl2e;
                    const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + one;
                    const f32x2 o = v * (f32x2){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                    acc[f][j][r] = o[0];
                    acc[f][j][r + 1] = o[1];
                }
    };
    auto keep = [&]() {
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int j = 0; j < 4; j++) xres[f][j] = acc[f][j];
    };
    auto to_image = [&](int layer) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float m = 0.f;
#pragma unroll
            for (int f = 0; f < 4; f++)
#pragma unroll
                for (int r = 0; r < 4; r++) m = fmaxf(m, fabsf(acc[f][j][r]));
            if (live[j]) atomicMax(&s_amax[layer][cpos[j]], __float_as_uint(m));
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) range_scale(__uint_as_float(s_amax[layer][cpos[j]]), sc[j], isc[j]);
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (live[j]) {
                    f16x4 oh, ol;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float v = acc[f][j][r] * sc[j];
                        oh[r] = (_Float16)v;
                        ol[r] = (_Float16)(v - (float)oh[r]);
                    }
                    const int at = crow[j] + ch0 + 16 * f;
                    *reinterpret_cast<f16x4*>(hi + at) = oh;
                    *reinterpret_cast<f16x4*>(lo + at) = ol;
                }
        __syncthreads();
    };
    // ---- stem
    zero();
    load_bias(w.b_stem);
    conv_split16<9, 1, false>(w.w16_stem, hi, SROWS * SRS, brow, lane, fbase, acc, pre);
    activate(w.inv_stem, false);
    keep();
    w_prefetch16(w.w16_3[0], 36, lane, fbase, pre);
    to_image(0);
    // ---- 3 residual blocks
    for (int blk = 0; blk < 3; blk++) {
        zero();
        load_bias(w.b3[blk]);
        conv_split16<9, 4, true>(w.w16_3[blk], hi, SROWS * SRS, brow, lane, fbase, acc, pre);
        activate(w.inv3[blk], false);
        w_prefetch16(w.w16_1[blk], 4, lane, fbase, pre);
        to_image(1 + 2 * blk);
        zero();
        load_bias(w.b1[blk]);
        conv_split16<1, 4, true>(w.w16_1[blk], hi, SROWS * SRS, brow, lane, fbase, acc, pre);
        activate(w.inv1[blk], true);
        keep();
        if (blk < 2) {
            w_prefetch16(w.w16_3[blk + 1], 36, lane, fbase, pre);
        } else {
#pragma unroll
            for (int cs = 0; cs < 4; cs++) {   // the heads' convolution weights ride in the first-slice registers
                pre.h[cs] = w.w_head[(size_t)(cs * 2 + 0) * 64 + lane];
                pre.l[cs] = w.w_head[(size_t)(cs * 2 + 1) * 64 + lane];
            }
        }
        to_image(2 + 2 * blk);
    }
    float pw[2][7], vw[3], hbias[4], pbias[7], vbias[3];
#pragma unroll
    for (int r = 0; r < 4; r++) hbias[r] = w.b_head[r];
#pragma unroll
    for (int a = 0; a < 7; a++) pbias[a] = w.plin_b[a];
#pragma unroll
    for (int k = 0; k < 3; k++) vbias[k] = w.vlin_b[k];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int a = 0; a < 7; a++) pw[i][a] = (wave < SNS && lane + 64 * i < 126) ? w.plin_w[a * 126 + lane + 64 * i] : 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) vw[k] = (wave < SNS && lane < 42) ? w.vlin_w[k * 42 + lane] : 0.f;
    // ---- heads: the four 16-cell groups of a wave column are dealt to its two waves
#pragma unroll
    for (int j0 = 0; j0 < 2; j0++) {
        const int jj = wa * 2 + j0;
        int col = wb * 64 + jj * 16 + n16;
        const bool lv = col < SCELLS;
        if (!lv) col = 0;
        const int sp = col / 42, p = col % 42;
        const int b16 = scell(sp, p / 7, p % 7) * SRS + 8 * g;
        f32x4 a4 = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int cs = 0; cs < 4; cs++) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(hi + b16 + cs * 32);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(lo + b16 + cs * 32);
            SMFMA16(pre.h[cs], bh, a4);
            SMFMA16(pre.h[cs], bl, a4);
            SMFMA16(pre.l[cs], bh, a4);
        }
        if (g == 0 && lv) {
            float hsc, hisc;
            range_scale(__uint_as_float(s_amax[6][sp]), hsc, hisc);
#pragma unroll
            for (int r = 0; r < 4; r++) hb[r][col] = ssilu(a4[r] * (w.inv_head * hisc) + hbias[r]);
        }
    }
    __syncthreads();
    // ---- Linear + log-softmax + predict(): wave s, position s (as in k_cnn_forward_split)
    const int s = wave;
    if (s >= SNS || s0 + s >= n) return;
    float pl[7], vl[3];
    {
        const float x0 = hb[lane / 42][s * 42 + lane % 42];
        const float x1 = lane + 64 < 126 ? hb[(lane + 64) / 42][s * 42 + (lane + 64) % 42] : 0.f;
        const float xv = lane < 42 ? hb[3][s * 42 + lane] : 0.f;
#pragma unroll
        for (int a = 0; a < 7; a++) pl[a] = swave_sum(x0 * pw[0][a] + x1 * pw[1][a]) + pbias[a];
#pragma unroll
        for (int k = 0; k < 3; k++) vl[k] = swave_sum(xv * vw[k]) + vbias[k];
    }
    float mx = pl[0];
#pragma unroll
    for (int a = 1; a < 7; a++) mx = fmaxf(mx, pl[a]);
    float se = 0.f;
#pragma unroll
    for (int a = 0; a < 7; a++) se += __expf(pl[a] - mx);
    const float lse = logf(se);
    const float mv = fmaxf(vl[0], fmaxf(vl[1], vl[2]));
    const float lsv = logf(__expf(vl[0] - mv) + __expf(vl[1] - mv) + __expf(vl[2] - mv));
    const i64 slot = s_slot[s];
    if (lane < 7) {
        float mine_l = pl[0];
#pragma unroll
        for (int a = 1; a < 7; a++) mine_l = lane == a ? pl[a] : mine_l;
        const float lp = mine_l - mx - lse;
        if (logp) logp[slot * 7 + lane] = lp;
        if (probs) probs[slot * 7 + lane] = __expf(lp);
    } else if (lane >= 8 && lane < 11) {
        const int k = lane - 8;
        const float lv = (k == 0 ? vl[0] : (k == 1 ? vl[1] : vl[2])) - mv - lsv;
        if (logv) logv[slot * 3 + k] = lv;
    } else if (lane == 16) {
        const float v0 = __expf(vl[0] - mv - lsv), v1 = __expf(vl[1] - mv - lsv), v2 = __expf(vl[2] - mv - lsv);
        if (value) value[slot] = (-s_player[s]) * 0.5f * v0 + v1 - v2;
    }
}

// ------------------------------------------------------------------------------------------------ host side
template <class T>
static int uploads(CnnS* c, const std::vector<T>& v, const T** out) {
    void* q = nullptr;
    AZ_HIP(hipMalloc(&q, v.size() * sizeof(T)));
    c->allocs.push_back(q);
    AZ_HIP(hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)q;
    return AZ_OK;
}

// One layer's folded weights (float64) -> power-of-two scale, fp16 hi / lo fragments.  `at(frag, lane, i)` gives the
// folded weight of fragment `frag` (hi and lo copies follow each other), or 0.
template <class F>
static float pack_split(int n_frag, F at, std::vector<_Float16>& out) {
    double mx = 0;
    for (int fr = 0; fr < n_frag; fr++)
        for (int l = 0; l < 64; l++)
            for (int i = 0; i < 8; i++) mx = fmax(mx, fabs(at(fr, l, i)));
    const double scale = exp2(floor(log2(16384.0 / (mx > 0 ? mx : 1.0))));
    out.assign((size_t)n_frag * 2 * 64 * 8, (_Float16)0.f);
    for (int fr = 0; fr < n_frag; fr++)
        for (int l = 0; l < 64; l++)
            for (int i = 0; i < 8; i++) {
                const float v = (float)(at(fr, l, i) * scale);
                const _Float16 h = (_Float16)v;
                out[(((size_t)fr * 2 + 0) * 64 + l) * 8 + i] = h;
                out[(((size_t)fr * 2 + 1) * 64 + l) * 8 + i] = (_Float16)(v - (float)h);
            }
    return (float)(1.0 / scale);
}

int cnns_create(const float* blob, size_t n_floats, CnnS** out) {
    if (!blob || n_floats != 500106) {
        set_error("az_net_load: expected 500106 float32 values (the reference CNN state_dict, see include/azamd.h)");
        return AZ_ERR_ARG;
    }
    CnnS* c = new CnnS();
    memset(&c->d, 0, sizeof(c->d));
    CnnBlobReader br{blob};
    int rc = AZ_OK;
    std::vector<_Float16> pk;
    auto up16 = [&](const f16x8** dst) {
        const _Float16* p = nullptr;
        if (rc == AZ_OK) rc = uploads(c, pk, &p);
        *dst = reinterpret_cast<const f16x8*>(p);
    };
    auto bias_of = [](const CnnBn& b, int n) {
        std::vector<float> v(n);
        for (int i = 0; i < n; i++) v[i] = cnn_bn_bias(b, i);
        return v;
    };
    const float* wt = br.take(128 * 3 * 9);
    CnnBn bn = cnn_take_bn(br, 128);
    c->d.inv_stem = pack_split(9 * 2 * SMFT, [&](int fr, int l, int i) {
        const int tap = fr / (2 * SMFT), kh = (fr / SMFT) % 2, f = fr % SMFT, co = 32 * f + l % 32, ci = 16 * kh + 8 * (l / 32) + i;
        return ci < 3 ? (double)wt[((size_t)co * 3 + ci) * 9 + tap] * (double)cnn_bn_scale(bn, co) : 0.0;
    }, pk);
    up16(&c->d.w_stem);
    const float inv_stem16 = pack_split(9 * 8, [&](int fr, int l, int i) {
        const int tap = fr / 8, f = fr % 8, co = 16 * f + l % 16, ci = 8 * (l / 16) + i;
        return ci < 3 ? (double)wt[((size_t)co * 3 + ci) * 9 + tap] * (double)cnn_bn_scale(bn, co) : 0.0;
    }, pk);
    up16(&c->d.w16_stem);
    if (rc == AZ_OK && inv_stem16 != c->d.inv_stem) rc = AZ_ERR_STATE;   // same weights, same power-of-two scale
    if (rc == AZ_OK) rc = uploads(c, bias_of(bn, 128), &c->d.b_stem);
    for (int b = 0; b < 3; b++) {
        wt = br.take((size_t)128 * 128 * 9);
        bn = cnn_take_bn(br, 128);
        c->d.inv3[b] = pack_split(36 * 2 * SMFT, [&](int fr, int l, int i) {
            const int sl = fr / (2 * SMFT), kh = (fr / SMFT) % 2, f = fr % SMFT, tap = sl >> 2, cs = sl & 3, co = 32 * f + l % 32,
                      ci = 32 * cs + 16 * kh + 8 * (l / 32) + i;
            return (double)wt[((size_t)co * SCH + ci) * 9 + tap] * (double)cnn_bn_scale(bn, co);
        }, pk);
        up16(&c->d.w3[b]);
        (void)pack_split(36 * 8, [&](int fr, int l, int i) {
            const int sl = fr / 8, f = fr % 8, tap = sl >> 2, cs = sl & 3, co = 16 * f + l % 16, ci = 32 * cs + 8 * (l / 16) + i;
            return (double)wt[((size_t)co * SCH + ci) * 9 + tap] * (double)cnn_bn_scale(bn, co);
        }, pk);
        up16(&c->d.w16_3[b]);
        if (rc == AZ_OK) rc = uploads(c, bias_of(bn, 128), &c->d.b3[b]);
        wt = br.take((size_t)128 * 128);
        bn = cnn_take_bn(br, 128);
        c->d.inv1[b] = pack_split(4 * 2 * SMFT, [&](int fr, int l, int i) {
            const int cc = fr / (2 * SMFT), kh = (fr / SMFT) % 2, f = fr % SMFT, co = 32 * f + l % 32,
                      ci = 32 * cc + 16 * kh + 8 * (l / 32) + i;
            return (double)wt[(size_t)co * SCH + ci] * (double)cnn_bn_scale(bn, co);
        }, pk);
        up16(&c->d.w1[b]);
        (void)pack_split(4 * 8, [&](int fr, int l, int i) {
            const int cc = fr / 8, f = fr % 8, co = 16 * f + l % 16, ci = 32 * cc + 8 * (l / 16) + i;
            return (double)wt[(size_t)co * SCH + ci] * (double)cnn_bn_scale(bn, co);
        }, pk);
        up16(&c->d.w16_1[b]);
        if (rc == AZ_OK) rc = uploads(c, bias_of(bn, 128), &c->d.b1[b]);
    }
    const float* pconv = br.take(3 * 128);
    CnnBn pbn = cnn_take_bn(br, 3);
    const float* plw = br.take(7 * 126);
    const float* plb = br.take(7);
    const float* vconv = br.take(128);
    CnnBn vbn = cnn_take_bn(br, 1);
    const float* vlw = br.take(3 * 42);
    const float* vlb = br.take(3);
    c->d.inv_head = pack_split(4, [&](int cs, int l, int i) {
        const int row = l % 16, ci = 32 * cs + 8 * (l / 16) + i;
        if (row < 3) return (double)pconv[row * 128 + ci] * (double)cnn_bn_scale(pbn, row);
        if (row == 3) return (double)vconv[ci] * (double)cnn_bn_scale(vbn, 0);
        return 0.0;
    }, pk);
    up16(&c->d.w_head);
    std::vector<float> bh = {cnn_bn_bias(pbn, 0), cnn_bn_bias(pbn, 1), cnn_bn_bias(pbn, 2), cnn_bn_bias(vbn, 0)};
    if (rc == AZ_OK) rc = uploads(c, bh, &c->d.b_head);
    if (rc == AZ_OK) rc = uploads(c, std::vector<float>(plw, plw + 7 * 126), &c->d.plin_w);
    if (rc == AZ_OK) rc = uploads(c, std::vector<float>(plb, plb + 7), &c->d.plin_b);
    if (rc == AZ_OK) rc = uploads(c, std::vector<float>(vlw, vlw + 3 * 42), &c->d.vlin_w);
    if (rc == AZ_OK) rc = uploads(c, std::vector<float>(vlb, vlb + 3), &c->d.vlin_b);
    if (rc == AZ_OK && (hipFuncSetAttribute((const void*)k_cnn_forward_split<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)split_lds(4)) != hipSuccess ||
                        hipFuncSetAttribute((const void*)k_cnn_forward_split<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)split_lds(4)) != hipSuccess ||
                        hipFuncSetAttribute((const void*)k_cnn_forward_split<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)split_lds(2)) != hipSuccess ||
                        hipFuncSetAttribute((const void*)k_cnn_forward_split<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)split_lds(3)) != hipSuccess ||
                        hipFuncSetAttribute((const void*)k_cnn_forward_split_strided<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)split_lds(3)) != hipSuccess ||
                        hipFuncSetAttribute((const void*)k_cnn_forward_split_strided<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)split_lds(2)) != hipSuccess ||
                        hipFuncSetAttribute((const void*)k_cnn_forward_split16, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)split_lds(3)) != hipSuccess)) {
        set_error("hipFuncSetAttribute(k_cnn_forward_split, dynamic LDS) failed");
        rc = AZ_ERR_HIP;
    }
    if (rc != AZ_OK) {
        cnns_destroy(c);
        return rc;
    }
    *out = c;
    return AZ_OK;
}

#ifdef AZ_K3S_DBG_TS
extern "C" int az_dbg_split_stamps(unsigned long long* out /* [4][64] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_split_ts), sizeof(unsigned long long) * 4 * 64) == hipSuccess ? 0 : -1;
}
#endif

void cnns_destroy(CnnS* c) {
    if (!c) return;
    for (void* q : c->allocs) (void)hipFree(q);
    delete c;
}

int cnns_launch(CnnS* c, unsigned max_n, i64 n_host, const u32* n_ptr, const u32* list, const u64* states, float* logp,
                float* logv, float* probs, float* value, hipStream_t st, unsigned grid_cap) {
    if (max_n == 0) return AZ_OK;
    // a launch whose capacity (max_n) is far above what it usually holds: a grid of grid_cap workgroups that stride over the batch
    static const char* stride_env = getenv("AZ_K3_STRIDE");   // "0": off (A/B)
    if (grid_cap && n_ptr && !logp && !logv && !getenv("AZ_K3_SPLIT_CFG") && !(stride_env && atoi(stride_env) == 0)) {
        if (grid_cap <= 32 && (max_n + 1) / 2 > grid_cap) {   // a handful of games: the two-position form (one workgroup alone: 52 us)
            k_cnn_forward_split_strided<2, 4><<<dim3(grid_cap), dim3(256), split_lds(2), st>>>(c->d, n_ptr, list, states, probs, value, max_n);
            AZ_HIP(hipGetLastError());
            return AZ_OK;
        }
        if (grid_cap > 32 && (max_n + 2) / 3 > grid_cap) {
            k_cnn_forward_split_strided<3, 4><<<dim3(grid_cap), dim3(256), split_lds(3), st>>>(c->d, n_ptr, list, states, probs, value, max_n);
            AZ_HIP(hipGetLastError());
            return AZ_OK;
        }
    }
    static const char* cfg = getenv("AZ_K3_SPLIT_CFG");   // tuning hook: "44" (4 positions, 4 waves), "48", "24"
    // A launch of a few leaves (the drop-in player's single game: one leaf) is ONE workgroup's latency whatever the form, and the
    // two-position form's workgroup is the shortest (one workgroup alone: 52 us against 65 us for <3,4> and 95 us for <4,4>).  All
    // forms deliver the same bits (tests/test_gpu_cnn.py::test_split_forms_deliver_the_same_bits), so what a launch's size
    // selects never shows in a result.
    const int kc = cfg ? atoi(cfg) : (max_n <= 64 ? 24 : SPLIT_DEFAULT_CFG);
    static const char* dsy = getenv("AZ_K3_SPLIT_DESYNC");
    const int desync = dsy ? atoi(dsy) : 0;
    if (kc == 316)   // the product configuration on v_mfma_f32_16x16x32_f16 (A/B form)
        k_cnn_forward_split16<<<dim3((max_n + 2) / 3), dim3(256), split_lds(3), st>>>(c->d, n_host, n_ptr, list, states, logp, logv, probs, value, max_n);
    else if (kc == 48)
        k_cnn_forward_split<4, 8><<<dim3((max_n + 3) / 4), dim3(512), split_lds(4), st>>>(c->d, n_host, n_ptr, list, states, logp, logv, probs, value, desync, max_n);
    else if (kc == 34)
        k_cnn_forward_split<3, 4><<<dim3((max_n + 2) / 3), dim3(256), split_lds(3), st>>>(c->d, n_host, n_ptr, list, states, logp, logv, probs, value, desync, max_n);
    else if (kc == 24)
        k_cnn_forward_split<2, 4><<<dim3((max_n + 1) / 2), dim3(256), split_lds(2), st>>>(c->d, n_host, n_ptr, list, states, logp, logv, probs, value, desync, max_n);
    else
        k_cnn_forward_split<4, 4><<<dim3((max_n + 3) / 4), dim3(256), split_lds(4), st>>>(c->d, n_host, n_ptr, list, states, logp, logv, probs, value, desync, max_n);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

}  // namespace az
