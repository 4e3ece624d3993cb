// az_cnn_wino.hip -- K3, Winograd form: the Connect4 policy/value CNN as one fused kernel per leaf batch, with the three
// 128->128 3x3 convolutions computed as Winograd F(2x2,3x3) products on fp32 MFMA (gfx950).
//
// Same function as az_cnn.hip's direct kernel (CNN.forward / CNN.predict, environments/Connect4/Network.py:16-27,60-64,
// 78-85), same 1e-4 contract, 1.97x fewer multiply-adds in the 3x3 layers:
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A   per 2x2 output tile, 4x4 input patch d, 16 transform points
//   6x7 board -> 3 x 4 tiles (output column 7 is computed and dropped): 12 x 16 = 192 products per (ci, co) vs 42 x 9 = 378.
// G g G^T (with the BatchNorm scale folded) is computed once on the host in float64; B^T d B is computed by each lane
// for its own (ci, tile) on the fly from 16 LDS reads (32 adds), so the transformed input never exists in memory;
// A^T M A runs on the accumulators in registers.
//
// Mapping to CDNA4.  One workgroup = 5 samples = 60 tile columns, 4 waves (one per SIMD), wave w owns columns
// 16w..16w+15 (wave 3: 12 real ones).  v_mfma_f32_16x16x4_f32: D[co 16][col 16] += U_p[co][ci 4] * V_p[ci 4][col].
//   * a pass = one M-tile (16 output channels): 32 k-steps x 16 points; 16 accumulator tiles (64 registers);
//   * after a pass the output transform + BN bias + SiLU give h[16 co][col][2x2] in registers, and those registers ARE
//     the B operands of the 1x1 convolution (k-step = accumulator register index r: lane group g holds co 4g + r),
//     accumulated over the 8 passes into 32 tiles (128 registers): the hidden tensor of a block never touches LDS;
//   * activations: LDS image [ci][cell], board rows padded to 9 cells [0, c0..c6, 0] so that a patch row is 4 contiguous
//     cells with its out-of-board columns reading zeros, plus one row of zeros for the patch rows above / below the board
//     (zero padding costs no instruction); 144 KB, written in place by the residual epilogue;
//   * weights: pre-swizzled per-lane A operand streams, 16-byte loads, L2-resident (3.4 MB), the 4 waves of a workgroup
//     walk the same stream so it is served by the CU's vector L1.
#include <math.h>
#include <string.h>

#include <vector>

#include "az_cnn.hpp"

namespace az {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 128;
constexpr int NPOS = 42;
constexpr int WS = 5;             // samples per workgroup
constexpr int WCOLS = WS * 12;    // tile columns per workgroup
constexpr int WP = WS * NPOS;     // 210 positions
// LDS activation image [ci][cell]: a board row is 9 cells [0, c0..c6, 0] (so columns -1, 7 and 8 = the next row's pad
// are zeros and a patch row is 4 CONTIGUOUS cells), a sample 6 such rows, then a 10-cell row of zeros that the patch
// rows above / below the board point to, then a dump cell for the outputs that do not exist.
constexpr int WSR = 9;            // cells per board row
constexpr int WSS = 6 * WSR;      // cells per sample
constexpr int WZ = WS * WSS;      // 270: the zero row
constexpr int WDUMP = WZ + 10;    // 280
constexpr int WRS = 281;          // row stride (floats)
__host__ __device__ constexpr int wcell(int s, int row, int col) { return s * WSS + row * WSR + col + 1; }
constexpr int NKS = HD / 4;       // k-steps of 4 input channels
constexpr int MTP = 2;            // M-tiles (16 output channels each) per pass
constexpr int NPASS = 8 / MTP;

struct CnnWDev {
    const float4* w_stem;      // [9 tap][2 q][64 lane]: co = 16(4q+x) + n, plane g
    const float4* w3[3];       // [NPASS][32 ks][MTP mtl][4 q][64 lane]: point 4q+x of U[co = 16(pass MTP + mtl) + n][ci = 4ks + g]
    const float4* w1[3];       // [32 kk = mt*4 + r][2 q][64 lane]: W1[co2 = 16(4q+x) + n][ci = 16mt + 4g + r]
    const float* w_head;       // [32 kk][64 lane]: row n (0..2 policy, 3 value), ci = 16mt + 4g + r
    const float* b_stem;       // [128]
    const float* b3[3];        // [128]
    const float* b1[3];        // [128]
    const float* b_head;       // [4]
    const float* plin_w;       // [7][126]
    const float* plin_b;       // [7]
    const float* vlin_w;       // [3][42]
    const float* vlin_b;       // [3]
};

struct CnnW {
    CnnWDev d;
    std::vector<void*> allocs;
};

#ifdef AZ_K3_DBG_TS  // diagnostic: shader-clock stamps of wave 0 of workgroup 0 (tools/k3_stamps.py)
__device__ unsigned long long g_wts[512];
#define WTS(i)                                                                    \
    do {                                                                          \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_wts[(i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define WTS(i) \
    do {       \
    } while (0)
#endif

__device__ inline float wsilu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ inline float wwave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// B^T d B for one 4x4 patch (row-major d[4i + j]) -> V[4i + j]
__device__ inline void wino_in(const float (&d)[16], float (&v)[16]) {
    float t[16];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        t[0 + j] = d[0 + j] - d[8 + j];
        t[4 + j] = d[4 + j] + d[8 + j];
        t[8 + j] = d[8 + j] - d[4 + j];
        t[12 + j] = d[4 + j] - d[12 + j];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        v[4 * i + 0] = t[4 * i + 0] - t[4 * i + 2];
        v[4 * i + 1] = t[4 * i + 1] + t[4 * i + 2];
        v[4 * i + 2] = t[4 * i + 2] - t[4 * i + 1];
        v[4 * i + 3] = t[4 * i + 1] - t[4 * i + 3];
    }
}

template <int U>
__device__ inline void wload4(float4 (&w)[U], const float4* p) {
#pragma unroll
    for (int q = 0; q < U; q++) w[q] = p[q * 64];
}
// xo = BYTE offsets into the activation image (saves the per-read index shift), koff in floats
__device__ inline float wlds(const float* xs, int byte_off, int koff) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xs) + byte_off + koff * 4);
}
// xo[i] = byte offset of patch row i (4 contiguous cells)
__device__ inline void wpatch(float (&d)[16], const float* xs, const int (&xo)[4], int koff) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // one 16-byte LDS read per patch row at 4-byte alignment (unaligned DS access mode of the HSA runtime)
        const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xs) + xo[i] + koff * 4);
        d[4 * i + 0] = v[0];
        d[4 * i + 1] = v[1];
        d[4 * i + 2] = v[2];
        d[4 * i + 3] = v[3];
    }
}

// Winograd-domain accumulate, D == C pinned in an accumulation register.  As a builtin, the register allocator (AGPR
// form: this kernel needs more than 256 registers) splits the 16 loop-carried tiles and rotates them with ~100
// v_accvgpr moves per trip; the tied asm operand keeps every tile in place.  The compiler pads no hazards of an asm
// statement: operands are produced a whole k-step before their use (wino_step), wmfma_enter() covers the accumulator
// initialisation, wmfma_drain() the readers of D.
__device__ inline void wmfma(f32x4& c, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// first k-step of a pass: D = A*B + 0, so that the 30 tiles without a bias need no initialisation at all
__device__ inline void wmfma0(f32x4& c, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
__device__ inline void wmfma_enter() { asm volatile("s_nop 3"); }
// 8-pass MFMA result -> any non-MFMA reader: 12 wait states.  Tiles 0..11 are at least 4 MFMAs (128 cycles) old here.
__device__ inline void wmfma_drain(f32x4 (&acc)[16 * MTP]) {
    asm volatile("s_nop 15" : "+a"(acc[16 * MTP - 4]), "+a"(acc[16 * MTP - 3]), "+a"(acc[16 * MTP - 2]),
                 "+a"(acc[16 * MTP - 1]));
}

__device__ inline void wfence8(const float4 (&w)[4 * MTP]) {
    if constexpr (MTP == 2)
        asm volatile("" ::"v"(w[0].x), "v"(w[1].x), "v"(w[2].x), "v"(w[3].x), "v"(w[4].x), "v"(w[5].x), "v"(w[6].x), "v"(w[7].x));
    else
        asm volatile("" ::"v"(w[0].x), "v"(w[1].x), "v"(w[2].x), "v"(w[3].x));
}
__device__ inline void wfence2(const float4 (&w)[2]) { asm volatile("" ::"v"(w[0].x), "v"(w[1].x)); }
__device__ inline void wmfma_operands(float (&y)[4]) {
    asm volatile("s_nop 1" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
}
// 8-pass MFMA result -> the epilogue's reads of the 1x1 tiles
__device__ inline void wmfma_drain2(f32x4 (&a)[8][4]) {
    asm volatile("s_nop 15" : "+a"(a[4][3]), "+a"(a[5][3]), "+a"(a[6][3]), "+a"(a[7][3]));
}
__device__ inline void wfence16(const float (&d)[16]) {
    asm volatile("" ::"v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8]),
                 "v"(d[9]), "v"(d[10]), "v"(d[11]), "v"(d[12]), "v"(d[13]), "v"(d[14]), "v"(d[15]));
}

// add number I (0..31) of B^T d B.  I < 16: t = B^T d (rows), I >= 16: v = t B (columns).
template <int I>
__device__ inline void wino_in_op(const float (&d)[16], float (&t)[16], float (&v)[16]) {
    if constexpr (I < 16) {
        constexpr int i = I / 4, j = I % 4;
        t[I] = i == 0 ? d[j] - d[8 + j] : i == 1 ? d[4 + j] + d[8 + j] : i == 2 ? d[8 + j] - d[4 + j] : d[4 + j] - d[12 + j];
    } else {
        constexpr int idx = I - 16, i = idx / 4, c = idx % 4;
        v[idx] = c == 0 ? t[4 * i] - t[4 * i + 2] : c == 1 ? t[4 * i + 1] + t[4 * i + 2]
                 : c == 2 ? t[4 * i + 2] - t[4 * i + 1] : t[4 * i + 1] - t[4 * i + 3];
    }
}

// MFMA number P of a k-step: M-tile P / 16 of the pass, transform point P % 16.
template <int P, bool FIRST>
__device__ inline void wino_mfma(const float4 (&cw)[4 * MTP], const float (&cv)[16], f32x4 (&acc)[16 * MTP]) {
    constexpr int Q = P >> 2;
    const float a = (P & 3) == 0 ? cw[Q].x : (P & 3) == 1 ? cw[Q].y : (P & 3) == 2 ? cw[Q].z : cw[Q].w;
    if constexpr (FIRST && (P & 15) != 5) wmfma0(acc[P], a, cv[P & 15]);   // tile 5 = point (1,1) starts from the BN bias
    else wmfma(acc[P], a, cv[P & 15]);
}
template <int P, int END, bool FIRST>
struct WinoMfmas {
    static __device__ inline void run(const float4 (&cw)[4 * MTP], const float (&cv)[16], f32x4 (&acc)[16 * MTP]) {
        wino_mfma<P, FIRST>(cw, cv, acc);
        if constexpr (P + 1 < END) WinoMfmas<P + 1, END, FIRST>::run(cw, cv, acc);
    }
};
template <int I>
struct WinoInOps {
    static __device__ inline void run(const float (&d)[16], float (&t)[16], float (&v)[16]) {
        wino_in_op<I>(d, t, v);
        if constexpr (I + 1 < 32) WinoInOps<I + 1>::run(d, t, v);
    }
};

// k-step k of a pass, software-pipelined by hand (one wave per SIMD: nothing else hides latency).
// On gfx950 an fp32 MFMA and VALU work of the same wave do not overlap, and every MFMA -> VALU switch costs ~10 cycles
// on top of the instruction itself (tools/ubench/mfma_bank.hip: MFMA/VALU alternating 46 cycles per MFMA, the same VALU
// bunched 39, bare 32.5).  So a k-step is: WHEAD MFMAs | ONE bunch of everything else | the remaining MFMAs, where
// the bunch = weight loads of step k+1 into nw, LDS reads of patch d(k+2) into dl, transform of patch d(k+1) = dt into nv.
#ifndef AZ_K3_WHEAD
#define AZ_K3_WHEAD 8
#endif
constexpr int WHEAD = AZ_K3_WHEAD;
template <bool FIRST = false>
__device__ inline void wino_step(const float4 (&cw)[4 * MTP], const float (&cv)[16], float4 (&nw)[4 * MTP],
                                 const float4* wnext, const float (&dt)[16], float (&nv)[16], float (&dl)[16],
                                 const float* xs, const int (&xo)[4], int kl, f32x4 (&acc)[16 * MTP]) {
    // one wait for everything this step consumes (requested a whole step ago)
    wfence8(cw);
    wfence16(dt);
    WinoMfmas<0, WHEAD, FIRST>::run(cw, cv, acc);
    __builtin_amdgcn_sched_barrier(0);
#ifndef AZ_K3_DBG_NOW   // diagnostic builds only (tools/k3_variants.sh): no weight traffic, results are garbage
    wload4<4 * MTP>(nw, wnext);
#endif
#ifdef AZ_K3_DBG_NOB    // diagnostic: no LDS patch reads (values made opaque so that the transform stays in the loop)
#pragma unroll
    for (int e = 0; e < 16; e++) asm volatile("" : "+v"(dl[e]));
#else
    wpatch(dl, xs, xo, kl);
#endif
#ifndef AZ_K3_DBG_NOXF  // diagnostic: no input transform
    float t[16];
    WinoInOps<0>::run(dt, t, nv);
#endif
    __builtin_amdgcn_sched_barrier(0);
    WinoMfmas<WHEAD, 16 * MTP, FIRST>::run(cw, cv, acc);
    __builtin_amdgcn_sched_barrier(0);
}

// A^T M A for accumulator register r of the 16 point tiles -> the 2x2 outputs y[2a + b]
template <int B>
__device__ inline void wino_out(const f32x4 (&acc)[16 * MTP], int r, float (&y)[4]) {
    float s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        s0[j] = acc[B + 0 + j][r] + acc[B + 4 + j][r] + acc[B + 8 + j][r];
        s1[j] = acc[B + 4 + j][r] - acc[B + 8 + j][r] - acc[B + 12 + j][r];
    }
    y[0] = s0[0] + s0[1] + s0[2];
    y[1] = s0[1] - s0[2] - s0[3];
    y[2] = s1[0] + s1[1] + s1[2];
    y[3] = s1[1] - s1[2] - s1[3];
}

// Linear(126->7)/Linear(42->3) + log-softmax + predict() for sample s of this workgroup, one wave.
__device__ inline void wino_finish(const CnnWDev& w, const float* hb, int s, int lane, i64 slot, float player,
                                   float* __restrict__ logp, float* __restrict__ logv, float* __restrict__ probs,
                                   float* __restrict__ value) {
    float pl[7], vl[3];
#pragma unroll
    for (int a = 0; a < 7; a++) pl[a] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) vl[k] = 0.f;
    for (int j = lane; j < 126; j += 64) {  // nn.Flatten of NCHW: index c*42 + h*7 + w
        const int c = j / NPOS, q = j % NPOS;
        const float xv = hb[c * WRS + wcell(s, q / 7, q % 7)];
#pragma unroll
        for (int a = 0; a < 7; a++) pl[a] += xv * w.plin_w[a * 126 + j];
    }
    if (lane < NPOS) {
        const float xv = hb[3 * WRS + wcell(s, lane / 7, lane % 7)];
#pragma unroll
        for (int k = 0; k < 3; k++) vl[k] += xv * w.vlin_w[k * NPOS + lane];
    }
#pragma unroll
    for (int a = 0; a < 7; a++) pl[a] = wwave_sum(pl[a]) + w.plin_b[a];
#pragma unroll
    for (int k = 0; k < 3; k++) vl[k] = wwave_sum(vl[k]) + w.vlin_b[k];
    if (lane == 0) {
        float mx = pl[0];
#pragma unroll
        for (int a = 1; a < 7; a++) mx = fmaxf(mx, pl[a]);
        float se = 0.f;
#pragma unroll
        for (int a = 0; a < 7; a++) se += expf(pl[a] - mx);
        const float lse = logf(se);
        const float mv = fmaxf(vl[0], fmaxf(vl[1], vl[2]));
        const float sv = expf(vl[0] - mv) + expf(vl[1] - mv) + expf(vl[2] - mv);
        const float lsv = logf(sv);
        float vp[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float lv = vl[k] - mv - lsv;
            if (logv) logv[slot * 3 + k] = lv;
            vp[k] = expf(lv);
        }
#pragma unroll
        for (int a = 0; a < 7; a++) {
            const float lp = pl[a] - mx - lse;
            if (logp) logp[slot * 7 + a] = lp;
            if (probs) probs[slot * 7 + a] = expf(lp);
        }
        // Network.py:83-84: classes [draw, mover wins, mover loses]
        if (value) value[slot] = (-player) * 0.5f * vp[0] + vp[1] - vp[2];
    }
}

// n_ptr != null: the batch size is read from device memory (leaf compaction count) and `list` maps batch index ->
// slot; outputs/inputs are then indexed by slot.
__global__ __launch_bounds__(256, 1) void k_cnn_forward_wino(CnnWDev w, i64 n_host, const u32* __restrict__ n_ptr,
                                                             const u32* __restrict__ list,
                                                             const u64* __restrict__ states, float* __restrict__ logp,
                                                             float* __restrict__ logv, float* __restrict__ probs,
                                                             float* __restrict__ value) {
    __shared__ float smem[(HD + 8) * WRS + 64];
    float* const xs = smem;              // activations [ci][sample*42 + pos], 109 KB
    float* const x0 = smem + HD * WRS;   // input planes (4th plane = 0); reused for the head outputs [c][pos]; doubles
                                         // as the slack the patch look-ahead reads (8 rows behind xs)
    __shared__ float s_player[WS];
    __shared__ long long s_slot[WS];
    const i64 n = n_ptr ? (i64)(*n_ptr) : n_host;
    const i64 s0 = (i64)blockIdx.x * WS;
    if (s0 >= n) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, nn = lane & 15;
    WTS(230);

    // ---- input planes from bitboards (env.current_state, env_cython.pyx:143-149) ----
    if (tid < WP) {
        const int p = tid, s = p / NPOS, q = p % NPOS, row = q / 7, col = q % 7;
        float vx = 0.f, vo = 0.f, vt = 0.f;
        if (s0 + s < n) {
            const i64 slot = list ? (i64)list[s0 + s] : (s0 + s);
            const u64 w0 = states[slot * 2], w1 = states[slot * 2 + 1];
            const int bit = col * 7 + (5 - row);
            vx = (float)((w0 >> bit) & 1ull);
            vo = (float)((w1 >> bit) & 1ull);
            vt = (w1 >> 63) ? -1.f : 1.f;
            if (q == 0) {
                s_player[s] = vt;
                s_slot[s] = slot;
            }
        }
        x0[0 * WRS + p] = vx;
        x0[1 * WRS + p] = vo;
        x0[2 * WRS + p] = vt;
        x0[3 * WRS + p] = 0.f;
    }
    // the cells nobody writes: 2 pads per board row, the zero row (and the dump cell); they stay zero for the whole kernel
    for (int i = tid; i < HD * 71; i += 256) {
        const int ch = i / 71, k = i % 71;
        const int cell = k < 60 ? (k >> 1) * WSR + (k & 1) * (WSR - 1) : WZ + (k - 60);
        xs[ch * WRS + cell] = 0.f;
    }

    // ---- stem: conv3x3(3->128) + BN + SiLU (Network.py:33-36), direct, columns = positions ----
    float4 wst[18];   // the whole stem weight stream of this lane (requested before the first barrier, used by every tile)
#pragma unroll
    for (int i = 0; i < 18; i++) wst[i] = w.w_stem[i * 64 + lane];
    float4 bst[8];
#pragma unroll
    for (int mt = 0; mt < 8; mt++) bst[mt] = *reinterpret_cast<const float4*>(w.b_stem + 16 * mt + 4 * g);
    __syncthreads();
    for (int nt = wave; nt * 16 < WP; nt += 4) {
        const int pos = nt * 16 + nn;
        const bool pv = pos < WP;
        const int q = pos % NPOS, row = q / 7, col = q % 7;
        const int cell = wcell(pos / NPOS, row, col);
        f32x4 a8[8];
#pragma unroll
        for (int mt = 0; mt < 8; mt++) {
            const float4 b = bst[mt];
            a8[mt] = (f32x4){b.x, b.y, b.z, b.w};
        }
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int dr = tap / 3 - 1, dc = tap % 3 - 1;
            const int rr = row + dr, cc = col + dc;
            const bool ok = pv && rr >= 0 && rr < 6 && cc >= 0 && cc < 7;
            const float b = ok ? x0[g * WRS + pos + dr * 7 + dc] : 0.f;
            const float4 wA = wst[tap * 2 + 0], wB = wst[tap * 2 + 1];
            a8[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA.x, b, a8[0], 0, 0, 0);
            a8[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA.y, b, a8[1], 0, 0, 0);
            a8[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA.z, b, a8[2], 0, 0, 0);
            a8[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA.w, b, a8[3], 0, 0, 0);
            a8[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB.x, b, a8[4], 0, 0, 0);
            a8[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB.y, b, a8[5], 0, 0, 0);
            a8[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB.z, b, a8[6], 0, 0, 0);
            a8[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB.w, b, a8[7], 0, 0, 0);
        }
        if (pv) {
#pragma unroll
            for (int mt = 0; mt < 8; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) xs[(16 * mt + 4 * g + r) * WRS + cell] = wsilu(a8[mt][r]);
        }
    }

    // ---- per-lane Winograd geometry: tile column -> patch cell offsets, output positions ----
    const int col_id = wave * 16 + nn;
    const bool cvalid = col_id < WCOLS;
    int off[4];       // patch row i -> BYTE offset of its 4 contiguous cells for k-step 0 (channel row g), or the zero row's
    int opos[4];      // output (a, b) -> cell index, or the dump cell for an output that does not exist
    {
        const int s = col_id / 12, t = col_id % 12, tr = t / 4, tc = t % 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int rr = 2 * tr - 1 + i;
            const bool ok = cvalid && rr >= 0 && rr < 6;
            off[i] = (g * WRS + (ok ? wcell(s, rr, 2 * tc - 1) : WZ)) * 4;   // columns -1, 7, 8 are the rows' zero pads
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                opos[2 * a + b] = (cvalid && (2 * tc + b < 7)) ? wcell(s, 2 * tr + a, 2 * tc + b) : WDUMP;
            }
    }
    __syncthreads();

    WTS(0);
    // ---- 3 residual blocks (Network.py:16-27) ----
    f32x4 acc2[8][4];                    // 1x1 output: [co2 tile][2x2 cell], register r <-> co2 = 16 mt2 + 4g + r
    constexpr int WSTEP = 4 * MTP * 64;  // float4 per k-step of the weight stream
    float4 wa[4 * MTP], wb[4 * MTP];     // weights of the current / next k-step
    float da[16], db[16], va[16], vb[16];
    wload4<4 * MTP>(wa, w.w3[0] + lane); // later blocks: requested before the previous block's epilogue
#pragma unroll 1
    for (int blk = 0; blk < 3; blk++) {
#pragma unroll
        for (int mt2 = 0; mt2 < 8; mt2++) {
            const float4 b = *reinterpret_cast<const float4*>(w.b1[blk] + 16 * mt2 + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; c++) acc2[mt2][c] = (f32x4){b.x, b.y, b.z, b.w};
        }
        const float4* w3l = w.w3[blk] + lane;
        const float4* w1l = w.w1[blk] + lane;
        // operands of k-steps 0 and 1 of the first pass; every later pass finds them requested during the previous
        // pass's 1x1 section (the weights by the last k-step's look-ahead: the stream is contiguous across passes)
        wpatch(da, xs, off, 0);
        wpatch(db, xs, off, 4 * WRS);
        wino_in(da, va);
#pragma unroll 1
        for (int pass = 0; pass < NPASS; pass++) {
            // -- Winograd-domain products for 16 MTP output channels: acc[16 mtl + p] (+ BN bias through point (1,1)) --
            f32x4 acc[16 * MTP];   // written by the first k-step (wmfma0); only the bias tiles are initialised
#pragma unroll
            for (int mtl = 0; mtl < MTP; mtl++) {
                const float4 b = *reinterpret_cast<const float4*>(w.b3[blk] + 16 * (pass * MTP + mtl) + 4 * g);
                acc[16 * mtl + 5] = (f32x4){b.x, b.y, b.z, b.w};
            }
            const float4* wp = w3l + (size_t)pass * NKS * WSTEP;
            int xo[4];                   // patch row byte offsets, advanced four k-steps per trip
#pragma unroll
            for (int e = 0; e < 4; e++) xo[e] = off[e];
            wmfma_enter();
            WTS(1 + (blk * 8 + pass) * 4 + 0);
            // look-ahead past the end (patches: two k-steps into the arrays behind xs; weights: one k-step into the next
            // pass / the allocation's slack) is loaded and never used: unconditional loads keep addressing on immediates.
            wino_step<true>(wa, va, wb, wp + WSTEP, db, vb, da, xs, xo, 8 * WRS, acc);
            wino_step(wb, vb, wa, wp + 2 * WSTEP, da, va, db, xs, xo, 12 * WRS, acc);
            wino_step(wa, va, wb, wp + 3 * WSTEP, db, vb, da, xs, xo, 16 * WRS, acc);
            wp += 4 * WSTEP;
            wino_step(wb, vb, wa, wp, da, va, db, xs, xo, 20 * WRS, acc);
#pragma unroll
            for (int e = 0; e < 4; e++) xo[e] += 16 * WRS * 4;
#pragma unroll 1
            for (int ks = 4; ks < NKS; ks += 4) {
                wino_step(wa, va, wb, wp + WSTEP, db, vb, da, xs, xo, 8 * WRS, acc);
                wino_step(wb, vb, wa, wp + 2 * WSTEP, da, va, db, xs, xo, 12 * WRS, acc);
                wino_step(wa, va, wb, wp + 3 * WSTEP, db, vb, da, xs, xo, 16 * WRS, acc);
                wp += 4 * WSTEP;
                wino_step(wb, vb, wa, wp, da, va, db, xs, xo, 20 * WRS, acc);
#pragma unroll
                for (int e = 0; e < 4; e++) xo[e] += 16 * WRS * 4;
            }
            wmfma_drain(acc);
            WTS(1 + (blk * 8 + pass) * 4 + 1);
            wpatch(da, xs, off, 0);          // next pass, k-steps 0 and 1 (the image does not change inside a block)
            wpatch(db, xs, off, 4 * WRS);
            // -- output transform, SiLU, and the 1x1 product with these hidden channels as reduction index --
            float4 ua[2], ub[2];
            const float4* w1p = w1l + (size_t)(pass * MTP * 4) * 2 * 64;
            wload4<2>(ua, w1p);
#pragma unroll
            for (int kk = 0; kk < 4 * MTP; kk++) {
                float y[4];
                if (kk < 4) wino_out<0>(acc, kk & 3, y);
                else wino_out<16 * (MTP - 1)>(acc, kk & 3, y);
#pragma unroll
                for (int c = 0; c < 4; c++) y[c] = wsilu(y[c]);
                float4(&cu)[2] = (kk & 1) ? ub : ua;
                float4(&nu)[2] = (kk & 1) ? ua : ub;
                wload4<2>(nu, w1p + (size_t)(kk + 1 < 4 * MTP ? kk + 1 : kk) * 2 * 64);
                wfence2(cu);          // one wait for this group's weights
                wmfma_operands(y);    // VALU result -> MFMA operand wait states (the asm MFMAs are not padded)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    wmfma(acc2[0][c], cu[0].x, y[c]);
                    wmfma(acc2[1][c], cu[0].y, y[c]);
                    wmfma(acc2[2][c], cu[0].z, y[c]);
                    wmfma(acc2[3][c], cu[0].w, y[c]);
                    wmfma(acc2[4][c], cu[1].x, y[c]);
                    wmfma(acc2[5][c], cu[1].y, y[c]);
                    wmfma(acc2[6][c], cu[1].z, y[c]);
                    wmfma(acc2[7][c], cu[1].w, y[c]);
                }
            }
            wino_in(da, va);
        }
        wload4<4 * MTP>(wa, w.w3[blk < 2 ? blk + 1 : 2] + lane);   // first weights of the next block
        wmfma_drain2(acc2);
        WTS(200 + blk * 3);
        __syncthreads();  // every wave has finished READING xs for this block's 3x3 convolution
        WTS(200 + blk * 3 + 1);
#pragma unroll
        for (int mt2 = 0; mt2 < 8; mt2++) {
            // F.silu(x + conv(x)); BN bias already in acc2.  16 reads, 16 SiLUs, 16 writes per tile (one LDS wait each
            // way); cells that do not exist (output column 7, columns past the 60th) go to the row's dump cell.
            float xv[16];
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) xv[4 * c + r] = xs[(16 * mt2 + 4 * g + r) * WRS + opos[c]];
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    xv[4 * c + r] = wsilu(xv[4 * c + r] + acc2[mt2][c][r]);
                    acc2[mt2][c][r] = xv[4 * c + r];
                }
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) xs[(16 * mt2 + 4 * g + r) * WRS + opos[c]] = xv[4 * c + r];
        }
        __syncthreads();
        WTS(200 + blk * 3 + 2);
    }

    WTS(220);
    // ---- heads: conv1x1(128->3)+BN+SiLU and conv1x1(128->1)+BN+SiLU as one 16-row MFMA tile (rows 0..3 used) ----
    float* hb = x0;
    {
        f32x4 hacc[4];
#pragma unroll
        for (int c = 0; c < 4; c++) hacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* wl = w.w_head + lane;
#pragma unroll
        for (int mt2 = 0; mt2 < 8; mt2++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float a = wl[(size_t)(mt2 * 4 + r) * 64];
#pragma unroll
                for (int c = 0; c < 4; c++)
                    hacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc2[mt2][c][r], hacc[c], 0, 0, 0);
            }
        if (g == 0) {  // rows 0..3 live in registers 0..3 of lanes 0..15
#pragma unroll
            for (int c = 0; c < 4; c++)
            {   // outputs that do not exist land in the rows' dump cells
                int op = opos[c];
                asm volatile("" : "+v"(op));   // computed here, not hoisted to the kernel's start and carried in registers
#pragma unroll
                for (int r = 0; r < 4; r++) hb[r * WRS + op] = wsilu(hacc[c][r] + w.b_head[r]);
            }
        }
    }
    __syncthreads();

    // ---- Linear + log-softmax + predict(): wave w finishes samples w and w + 4 ----
    for (int s = wave; s < WS; s += 4)
        if (s0 + s < n) wino_finish(w, hb, s, lane, (i64)s_slot[s], s_player[s], logp, logv, probs, value);
    WTS(231);
}

// ------------------------------------------------------------------------------------------------ host side
template <class T>
static int uploadw(CnnW* c, const std::vector<T>& v, const T** out) {
    void* q = nullptr;
    AZ_HIP(hipMalloc(&q, v.size() * sizeof(T) + 8192));  // + look-ahead slack
    c->allocs.push_back(q);
    AZ_HIP(hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)q;
    return AZ_OK;
}

// U = G g G^T per (co, ci), BN scale folded, float64 then rounded once; laid out as the per-lane A operand stream.
static void pack_wino3(const float* wt, const CnnBn& bn, std::vector<float4>& out) {
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> U((size_t)16 * HD * HD);  // [p][co][ci]
    for (int co = 0; co < HD; co++) {
        const double sc = (double)cnn_bn_scale(bn, co);
        for (int ci = 0; ci < HD; ci++) {
            const float* gk = wt + ((size_t)co * HD + ci) * 9;
            double t[4][3];
            for (int i = 0; i < 4; i++)
                for (int b = 0; b < 3; b++) {
                    double s = 0;
                    for (int a = 0; a < 3; a++) s += Gm[i][a] * ((double)gk[a * 3 + b] * sc);
                    t[i][b] = s;
                }
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) {
                    double s = 0;
                    for (int b = 0; b < 3; b++) s += t[i][b] * Gm[j][b];
                    U[((size_t)(4 * i + j) * HD + co) * HD + ci] = (float)s;
                }
        }
    }
    out.assign((size_t)8 * NKS * 4 * 64, make_float4(0, 0, 0, 0));
    for (int mt = 0; mt < 8; mt++)
        for (int ks = 0; ks < NKS; ks++)
            for (int q = 0; q < 4; q++)
                for (int lane = 0; lane < 64; lane++) {
                    const int co = 16 * mt + (lane & 15), ci = 4 * ks + (lane >> 4);
                    float v[4];
                    for (int x = 0; x < 4; x++) v[x] = U[((size_t)(4 * q + x) * HD + co) * HD + ci];
                    const int pass = mt / MTP, mtl = mt % MTP;
                    out[((((size_t)pass * NKS + ks) * MTP + mtl) * 4 + q) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
                }
}

int cnnw_create(const float* blob, size_t n_floats, CnnW** out) {
    if (!blob || n_floats != 500106) {
        set_error("az_net_load: expected 500106 float32 values (the reference CNN state_dict, see include/azamd.h)");
        return AZ_ERR_ARG;
    }
    CnnW* c = new CnnW();
    memset(&c->d, 0, sizeof(c->d));
    CnnBlobReader br{blob};
    int rc = AZ_OK;
    std::vector<float4> v4;
    auto bias_of = [](const CnnBn& b) {
        std::vector<float> v(HD);
        for (int i = 0; i < HD; i++) v[i] = cnn_bn_bias(b, i);
        return v;
    };
    // hidden.0 / hidden.1: stem, direct
    const float* w = br.take(128 * 3 * 9);
    CnnBn bn = cnn_take_bn(br, 128);
    v4.assign((size_t)9 * 2 * 64, make_float4(0, 0, 0, 0));
    for (int tap = 0; tap < 9; tap++)
        for (int q = 0; q < 2; q++)
            for (int lane = 0; lane < 64; lane++) {
                const int pl = lane >> 4;
                float v[4] = {0, 0, 0, 0};
                if (pl < 3)
                    for (int x = 0; x < 4; x++) {
                        const int co = 16 * (4 * q + x) + (lane & 15);
                        v[x] = w[((size_t)co * 3 + pl) * 9 + tap] * cnn_bn_scale(bn, co);
                    }
                v4[((size_t)tap * 2 + q) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
    if (rc == AZ_OK) rc = uploadw(c, v4, &c->d.w_stem);
    if (rc == AZ_OK) rc = uploadw(c, bias_of(bn), &c->d.b_stem);
    for (int b = 0; b < 3; b++) {  // hidden.4/5/6
        w = br.take((size_t)128 * 128 * 9);
        bn = cnn_take_bn(br, 128);
        pack_wino3(w, bn, v4);
        if (rc == AZ_OK) rc = uploadw(c, v4, &c->d.w3[b]);
        if (rc == AZ_OK) rc = uploadw(c, bias_of(bn), &c->d.b3[b]);
        w = br.take((size_t)128 * 128);
        bn = cnn_take_bn(br, 128);
        v4.assign((size_t)32 * 2 * 64, make_float4(0, 0, 0, 0));
        for (int kk = 0; kk < 32; kk++)
            for (int q = 0; q < 2; q++)
                for (int lane = 0; lane < 64; lane++) {
                    const int ci = 16 * (kk >> 2) + 4 * (lane >> 4) + (kk & 3);
                    float v[4];
                    for (int x = 0; x < 4; x++) {
                        const int co2 = 16 * (4 * q + x) + (lane & 15);
                        v[x] = w[(size_t)co2 * HD + ci] * cnn_bn_scale(bn, co2);
                    }
                    v4[((size_t)kk * 2 + q) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
                }
        if (rc == AZ_OK) rc = uploadw(c, v4, &c->d.w1[b]);
        if (rc == AZ_OK) rc = uploadw(c, bias_of(bn), &c->d.b1[b]);
    }
    // heads
    const float* pconv = br.take(3 * 128);
    CnnBn pbn = cnn_take_bn(br, 3);
    const float* plw = br.take(7 * 126);
    const float* plb = br.take(7);
    const float* vconv = br.take(128);
    CnnBn vbn = cnn_take_bn(br, 1);
    const float* vlw = br.take(3 * 42);
    const float* vlb = br.take(3);
    std::vector<float> wh((size_t)32 * 64, 0.f);
    for (int kk = 0; kk < 32; kk++)
        for (int lane = 0; lane < 64; lane++) {
            const int ci = 16 * (kk >> 2) + 4 * (lane >> 4) + (kk & 3), row = lane & 15;
            float v = 0.f;
            if (row < 3) v = pconv[row * 128 + ci] * cnn_bn_scale(pbn, row);
            else if (row == 3) v = vconv[ci] * cnn_bn_scale(vbn, 0);
            wh[(size_t)kk * 64 + lane] = v;
        }
    if (rc == AZ_OK) rc = uploadw(c, wh, &c->d.w_head);
    std::vector<float> bh = {cnn_bn_bias(pbn, 0), cnn_bn_bias(pbn, 1), cnn_bn_bias(pbn, 2), cnn_bn_bias(vbn, 0)};
    if (rc == AZ_OK) rc = uploadw(c, bh, &c->d.b_head);
    if (rc == AZ_OK) rc = uploadw(c, std::vector<float>(plw, plw + 7 * 126), &c->d.plin_w);
    if (rc == AZ_OK) rc = uploadw(c, std::vector<float>(plb, plb + 7), &c->d.plin_b);
    if (rc == AZ_OK) rc = uploadw(c, std::vector<float>(vlw, vlw + 3 * 42), &c->d.vlin_w);
    if (rc == AZ_OK) rc = uploadw(c, std::vector<float>(vlb, vlb + 3), &c->d.vlin_b);
    if (rc != AZ_OK) {
        cnnw_destroy(c);
        return rc;
    }
    *out = c;
    return AZ_OK;
}

void cnnw_destroy(CnnW* c) {
    if (!c) return;
    for (void* q : c->allocs) (void)hipFree(q);
    delete c;
}

int cnnw_launch(CnnW* c, unsigned max_n, i64 n_host, const u32* n_ptr, const u32* list, const u64* states, float* logp,
                float* logv, float* probs, float* value, hipStream_t st) {
    const unsigned grid = (max_n + WS - 1) / WS;
    if (grid == 0) return AZ_OK;
    k_cnn_forward_wino<<<dim3(grid), dim3(256), 0, st>>>(c->d, n_host, n_ptr, list, states, logp, logv, probs, value);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

}  // namespace az

#ifdef AZ_K3_DBG_TS
extern "C" int az_dbg_read_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(az::g_wts), sizeof(unsigned long long) * n);
}
#endif
