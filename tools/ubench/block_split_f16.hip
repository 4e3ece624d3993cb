// Prototype for the round-2 K3 plan, second step: ONE WHOLE residual block of the Connect4
// network, y = SiLU(x + b2 + W1 . SiLU(b1 + W3 * x))   (Network.py Block, BatchNorm folded into W and b), 4 positions per
// workgroup, every fp32 product split over fp16 MFMA (hi*hi + hi*lo + lo*hi, fp32 accumulation):
//   * 3x3 convolution as in conv_split_f16.hip (activation image pre-split in LDS, operands double-buffered);
//   * the 1x1 convolution chained from the accumulators: the k order of its weights is chosen so that a lane's own 8
//     accumulator registers of channel fragments 2c and 2c+1 ARE its B operand for k-slice c (no LDS round trip);
//   * epilogue: residual read back from the split image (hi + lo), SiLU, split again, written over the image in place.
// `rounds` blocks are applied in a row (same weights) so that the write-back path is exercised and timed.
// Checked against a float64 evaluation on the host; reports cycles per block.  Not part of the library.
//   build: hipcc --offload-arch=gfx950 -O3 -o dbgbuild/block_split_f16 tools/ubench/block_split_f16.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int CH = 128, NS = 4;
constexpr int ROWS = NS * 63 + 9 + 20, RS = CH + 8, FRONT = 10;
constexpr int NSL = 36, MF = 8, NF = 3, ROUNDS = 3;
__host__ __device__ inline int cell_row(int s, int r, int c) { return FRONT + s * 63 + (r + 1) * 9 + (c + 1); }
__host__ __device__ inline int chain_k(int c, int kg, int i) {  // channel that k-slice c holds at (k-group kg, element i)
    return 32 * c + (i < 4 ? 4 * kg + i : 16 + 4 * kg + (i - 4));
}

__device__ inline float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }  // v_exp_f32 + v_rcp_f32
__device__ inline void split8(const f32x4& a, const f32x4& b, f16x8& h, f16x8& l) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        h[i] = (_Float16)a[i];
        l[i] = (_Float16)(a[i] - (float)h[i]);
        h[4 + i] = (_Float16)b[i];
        l[4 + i] = (_Float16)(b[i] - (float)h[4 + i]);
    }
}

// w3: [slice][mfrag][hi|lo][lane] f16x8 (k = 32 channels of one tap); w1: [kslice][mfrag][hi|lo][lane] f16x8 (chain k order)
__global__ __launch_bounds__(256, 1) void k_block(const f16x8* __restrict__ w3, const f16x8* __restrict__ w1,
                                                  const float* __restrict__ b1, const float* __restrict__ b2,
                                                  const float* __restrict__ x, float* __restrict__ y, float inv_s3, float inv_s1,
                                                  unsigned long long* cyc, unsigned long long* phase) {
    extern __shared__ _Float16 lds[];
    _Float16* hi = lds;
    _Float16* lo = lds + ROWS * RS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < ROWS * RS; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0u;
    __syncthreads();
    const float* xb = x + (size_t)blockIdx.x * NS * CH * 42;
    for (int i = tid; i < NS * CH * 42; i += 256) {
        const int s = i / (CH * 42), ci = (i / 42) % CH, p = i % 42;
        const float v = xb[i];
        const _Float16 h = (_Float16)v;
        const int row = cell_row(s, p / 7, p % 7);
        hi[row * RS + ci] = h;
        lo[row * RS + ci] = (_Float16)(v - (float)h);
    }
    __syncthreads();
    int brow[NF], crow[NF];
    bool live[NF];
#pragma unroll
    for (int j = 0; j < NF; j++) {
        int col = wave * 48 + j * 16 + n16;
        live[j] = col < NS * 42;
        if (!live[j]) col = 0;
        const int s = col / 42, p = col % 42;
        crow[j] = cell_row(s, p / 7, p % 7) * RS;
        brow[j] = crow[j] + 8 * g;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tp = t0;
#define STAMP(i) { const unsigned long long tn_ = __builtin_readcyclecounter(); ph[i] += tn_ - tp; tp = tn_; }
    for (int round = 0; round < ROUNDS; round++) {
        // the real kernel has other weights in every block: keep the compiler from hoisting these rounds' common loads
        asm volatile("" : "+s"(w1), "+s"(w3), "+s"(b1), "+s"(b2));
        f32x4 acc[MF][NF];
#pragma unroll
        for (int f = 0; f < MF; f++)
#pragma unroll
            for (int j = 0; j < NF; j++) acc[f][j] = (f32x4){0, 0, 0, 0};
        {  // ---- 3x3 convolution
            f16x8 ah[2][MF], al[2][MF], bh[2][NF], bl[2][NF];
            auto fetch = [&](int sl, int b) {
                const int tap = sl >> 2, cs = sl & 3;
                const int toff = ((tap / 3 - 1) * 9 + (tap % 3 - 1)) * RS + cs * 32;
#pragma unroll
                for (int f = 0; f < MF; f++) {
                    ah[b][f] = w3[((size_t)(sl * MF + f) * 2 + 0) * 64 + lane];
                    al[b][f] = w3[((size_t)(sl * MF + f) * 2 + 1) * 64 + lane];
                }
#pragma unroll
                for (int j = 0; j < NF; j++) {
                    bh[b][j] = *reinterpret_cast<const f16x8*>(hi + brow[j] + toff);
                    bl[b][j] = *reinterpret_cast<const f16x8*>(lo + brow[j] + toff);
                }
            };
            auto mfmas = [&](int b) {
#pragma unroll
                for (int f = 0; f < MF; f++)
#pragma unroll
                    for (int j = 0; j < NF; j++) {
                        acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b][f], bh[b][j], acc[f][j], 0, 0, 0);
                        acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b][f], bl[b][j], acc[f][j], 0, 0, 0);
                        acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b][f], bh[b][j], acc[f][j], 0, 0, 0);
                    }
            };
            fetch(0, 0);
            for (int sl = 0; sl < NSL; sl += 2) {
                fetch(sl + 1, 1);
                mfmas(0);
                if (sl + 2 < NSL) fetch(sl + 2, 0);
                mfmas(1);
            }
        }
        STAMP(0);
        // ---- 1x1 convolution chained from the registers
        f32x4 acc2[MF][NF];
#pragma unroll
        for (int f = 0; f < MF; f++)
#pragma unroll
            for (int j = 0; j < NF; j++) acc2[f][j] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            f16x8 th[NF], tl[NF], ah[MF], al[MF];
#pragma unroll
            for (int f = 0; f < MF; f++) {
                ah[f] = w1[((size_t)(c * MF + f) * 2 + 0) * 64 + lane];
                al[f] = w1[((size_t)(c * MF + f) * 2 + 1) * 64 + lane];
            }
            {  // t = SiLU(conv + b1) for the 32 channels of this k-slice, straight into the split operands
                const f32x4 ba = *reinterpret_cast<const f32x4*>(b1 + 32 * c + 4 * g);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b1 + 32 * c + 16 + 4 * g);
#pragma unroll
                for (int j = 0; j < NF; j++) {
                    f32x4 ta, tb;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        ta[r] = silu(acc[2 * c][j][r] * inv_s3 + ba[r]);
                        tb[r] = silu(acc[2 * c + 1][j][r] * inv_s3 + bb[r]);
                    }
                    split8(ta, tb, th[j], tl[j]);
                }
            }
#pragma unroll
            for (int f = 0; f < MF; f++)
#pragma unroll
                for (int j = 0; j < NF; j++) {
                    acc2[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], th[j], acc2[f][j], 0, 0, 0);
                    acc2[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], tl[j], acc2[f][j], 0, 0, 0);
                    acc2[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[f], th[j], acc2[f][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);  // keep the next slice's 16 weight fragments from being hoisted up here
        }
        STAMP(1);
        // ---- out = SiLU(x + conv1x1 + b2): residual from the split image (a lane reads only the cells and channels it
        // is about to overwrite), result split again and written in place once every wave is done with the old image
        __syncthreads();
        STAMP(2);
#pragma unroll
        for (int f = 0; f < MF; f++) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + 16 * f + 4 * g);
#pragma unroll
            for (int j = 0; j < NF; j++) {
                const int at = crow[j] + 16 * f + 4 * g;
                const f16x4 xh = *reinterpret_cast<const f16x4*>(hi + at), xl = *reinterpret_cast<const f16x4*>(lo + at);
                f16x4 oh, ol;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v = silu(((float)xh[r] + (float)xl[r]) + (acc2[f][j][r] * inv_s1 + bb[r]));
                    oh[r] = (_Float16)v;
                    ol[r] = (_Float16)(v - (float)oh[r]);
                    acc2[f][j][r] = v;
                }
                if (live[j]) {
                    *reinterpret_cast<f16x4*>(hi + at) = oh;
                    *reinterpret_cast<f16x4*>(lo + at) = ol;
                }
            }
        }
        STAMP(3);
        __syncthreads();
        STAMP(4);
        if (round == ROUNDS - 1) {
            float* yb = y + (size_t)blockIdx.x * NS * 42 * CH;
#pragma unroll
            for (int j = 0; j < NF; j++)
                if (live[j]) {
                    const int col = wave * 48 + j * 16 + n16;
#pragma unroll
                    for (int f = 0; f < MF; f++)
#pragma unroll
                        for (int r = 0; r < 4; r++) yb[col * CH + 16 * f + 4 * g + r] = acc2[f][j][r];
                }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0 && blockIdx.x == 0) {
        *cyc = t1 - t0;
        for (int i = 0; i < 5; i++) phase[i] = ph[i];
    }
}

#define CK(x)                                                     \
    do {                                                          \
        hipError_t e_ = (x);                                      \
        if (e_ != hipSuccess) {                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                             \
        }                                                         \
    } while (0)

static void pack(std::vector<_Float16>& pk, size_t at, float v) {
    const _Float16 h = (_Float16)v;
    pk[at] = h;
    pk[at + 64 * 8] = (_Float16)(v - (float)h);  // the lo fragment follows the hi fragment
}

int main() {
    const int WG = 256;
    srand(2);
    auto rnd = []() { return (rand() / (double)RAND_MAX) * 2.0 - 1.0; };
    std::vector<float> w3((size_t)CH * CH * 9), w1((size_t)CH * CH), b1(CH), b2(CH), x((size_t)WG * NS * CH * 42);
    for (auto& v : w3) v = (float)(0.05 * rnd() * rnd());
    for (auto& v : w1) v = (float)(0.12 * rnd() * rnd());
    for (auto& v : b1) v = (float)(0.3 * rnd());
    for (auto& v : b2) v = (float)(0.3 * rnd());
    for (auto& v : x) v = (float)(1.5 * rnd());
    auto pow2_scale = [](const std::vector<float>& w) {
        float m = 0;
        for (auto v : w) m = fmaxf(m, fabsf(v));
        return exp2f(floorf(log2f(16384.0f / m)));
    };
    const float s3 = pow2_scale(w3), s1 = pow2_scale(w1);
    std::vector<_Float16> p3((size_t)NSL * MF * 2 * 64 * 8), p1((size_t)4 * MF * 2 * 64 * 8);
    for (int sl = 0; sl < NSL; sl++)
        for (int f = 0; f < MF; f++)
            for (int l = 0; l < 64; l++)
                for (int i = 0; i < 8; i++) {
                    const int tap = sl >> 2, cs = sl & 3, co = 16 * f + l % 16, ci = 32 * cs + 8 * (l / 16) + i;
                    pack(p3, ((((size_t)sl * MF + f) * 2) * 64 + l) * 8 + i, w3[((size_t)co * CH + ci) * 9 + tap] * s3);
                }
    for (int c = 0; c < 4; c++)
        for (int f = 0; f < MF; f++)
            for (int l = 0; l < 64; l++)
                for (int i = 0; i < 8; i++) {
                    const int co = 16 * f + l % 16, ci = chain_k(c, l / 16, i);
                    pack(p1, ((((size_t)c * MF + f) * 2) * 64 + l) * 8 + i, w1[(size_t)co * CH + ci] * s1);
                }
    f16x8 *d3, *d1;
    float *db1, *db2, *dx, *dy;
    unsigned long long *dc, *dph;
    CK(hipMalloc(&dph, 40));
    CK(hipMalloc(&d3, p3.size() * 2));
    CK(hipMalloc(&d1, p1.size() * 2));
    CK(hipMalloc(&db1, CH * 4));
    CK(hipMalloc(&db2, CH * 4));
    CK(hipMalloc(&dx, x.size() * 4));
    CK(hipMalloc(&dy, (size_t)WG * NS * 42 * CH * 4));
    CK(hipMalloc(&dc, 8));
    CK(hipMemcpy(d3, p3.data(), p3.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d1, p1.data(), p1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db1, b1.data(), CH * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db2, b2.data(), CH * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    const size_t smem = (size_t)2 * ROWS * RS * 2;
    CK(hipFuncSetAttribute((const void*)k_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        k_block<<<WG, 256, smem>>>(d3, d1, db1, db2, dx, dy, 1.0f / s3, 1.0f / s1, dc, dph);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipGetLastError());
    unsigned long long c = 0;
    CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    unsigned long long ph[5];
    CK(hipMemcpy(ph, dph, 40, hipMemcpyDeviceToHost));
    printf("phases per block (k cycles, wave 0): 3x3 conv %.1f | SiLU + split + chained 1x1 %.1f | barrier wait %.1f | residual + SiLU + split + store %.1f | barrier wait %.1f\n",
           ph[0] / 1e3 / ROUNDS, ph[1] / 1e3 / ROUNDS, ph[2] / 1e3 / ROUNDS, ph[3] / 1e3 / ROUNDS, ph[4] / 1e3 / ROUNDS);
    std::vector<float> yh((size_t)NS * 42 * CH);
    CK(hipMemcpy(yh.data(), dy, yh.size() * 4, hipMemcpyDeviceToHost));
    // float64 reference for workgroup 0: ROUNDS blocks in a row
    std::vector<double> cur((size_t)NS * CH * 42), t((size_t)CH * 42), nxt((size_t)NS * CH * 42);
    for (size_t i = 0; i < cur.size(); i++) cur[i] = x[i];
    auto sl64 = [](double v) { return v / (1.0 + exp(-v)); };
    for (int round = 0; round < ROUNDS; round++) {
        for (int s = 0; s < NS; s++) {
            for (int co = 0; co < CH; co++)
                for (int p = 0; p < 42; p++) {
                    double a = b1[co];
                    for (int ci = 0; ci < CH; ci++)
                        for (int k = 0; k < 9; k++) {
                            const int rr = p / 7 + k / 3 - 1, c2 = p % 7 + k % 3 - 1;
                            if (rr < 0 || rr >= 6 || c2 < 0 || c2 >= 7) continue;
                            a += cur[((size_t)s * CH + ci) * 42 + rr * 7 + c2] * w3[((size_t)co * CH + ci) * 9 + k];
                        }
                    t[(size_t)co * 42 + p] = sl64(a);
                }
            for (int co = 0; co < CH; co++)
                for (int p = 0; p < 42; p++) {
                    double a = b2[co];
                    for (int ci = 0; ci < CH; ci++) a += t[(size_t)ci * 42 + p] * w1[(size_t)co * CH + ci];
                    nxt[((size_t)s * CH + co) * 42 + p] = sl64(cur[((size_t)s * CH + co) * 42 + p] + a);
                }
        }
        cur = nxt;
    }
    double emax = 0, ymax = 0;
    for (int s = 0; s < NS; s++)
        for (int p = 0; p < 42; p++)
            for (int co = 0; co < CH; co++) {
                const double want = cur[((size_t)s * CH + co) * 42 + p], got = yh[(size_t)(s * 42 + p) * CH + co];
                emax = fmax(emax, fabs(got - want));
                ymax = fmax(ymax, fabs(want));
            }
    const double flop = 2.0 * CH * CH * 10 * 42 * NS * ROUNDS;
    printf("LDS %zu B, weight scales 2^%d / 2^%d, %d blocks in a row\n", smem, (int)log2f(s3), (int)log2f(s1), ROUNDS);
    printf("%llu cycles per workgroup of %d positions = %.1f k cycles per block (%.1f k per position and block)\n", c, NS,
           c / 1e3 / ROUNDS, c / 1e3 / ROUNDS / NS);
    printf("kernel %.3f ms for %d workgroups: %.1f TFLOP/s algorithmic fp32\n", ms, WG, flop * WG / ms / 1e9);
    printf("max |y| %.3f; max abs error vs float64 after %d blocks: %.3e\n", ymax, ROUNDS, emax);
    return emax < 1e-4 * ymax ? 0 : 2;
}
