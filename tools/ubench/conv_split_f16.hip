// Prototype for the round-2 K3 plan: ONE 3x3 convolution layer 128 -> 128 channels on a 6x7
// board, 4 positions per workgroup, computed as a direct implicit GEMM on v_mfma_f32_16x16x32_f16 with every fp32
// product split as hi*hi + hi*lo + lo*hi (fp16 halves, fp32 accumulation).  The activation image sits in LDS already
// split (hi plane, lo plane, channel-contiguous rows), the weights are split and packed on the host.  Checks the result
// against a float64 convolution on the host and reports cycles per workgroup.  Not part of the library.
//   build: hipcc --offload-arch=gfx950 -O3 -o dbgbuild/conv_split_f16 tools/ubench/conv_split_f16.hip ; run on the box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int CI = 128, CO = 128, NS = 4;           // channels, positions per workgroup
constexpr int ROWS = NS * 63 + 9 + 20;              // padded cells: per position 7 rows x 9 (shared zero row), + margins
constexpr int RS = CI + 8;                          // halves per LDS row (272 B: 16-byte reads of 16 lanes spread over banks)
constexpr int FRONT = 10;                           // margin so that tap offsets (-10..+10) stay inside
constexpr int NSL = 36;                             // k-slices of 32: 9 taps x 4 channel groups
constexpr int MF = 8, NF = 3;                       // wave tile: 8 x 16 output channels, 3 x 16 cells
__host__ __device__ inline int cell_row(int s, int r, int c) { return FRONT + s * 63 + (r + 1) * 9 + (c + 1); }

// weights: [slice][mfrag][hi|lo][lane] f16x8;  x: [NS][CI][42] fp32 per workgroup;  y: [168 cells][CO] fp32 per workgroup
__global__ __launch_bounds__(256, 1) void k_conv(const f16x8* __restrict__ wpk, const float* __restrict__ x, float* __restrict__ y,
                                                 float inv_scale, unsigned long long* cyc) {
    extern __shared__ _Float16 lds[];
    _Float16* hi = lds;
    _Float16* lo = lds + ROWS * RS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * ROWS * RS / 2; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0u;
    __syncthreads();
    const float* xb = x + (size_t)blockIdx.x * NS * CI * 42;
    for (int i = tid; i < NS * CI * 42; i += 256) {  // split the fp32 image once: what a layer's epilogue would do
        const int s = i / (CI * 42), ci = (i / 42) % CI, p = i % 42;
        const float v = xb[i];
        const _Float16 h = (_Float16)v;
        const int row = cell_row(s, p / 7, p % 7);
        hi[row * RS + ci] = h;
        lo[row * RS + ci] = (_Float16)(v - (float)h);
    }
    __syncthreads();
    // this wave's 48 cells (columns of the GEMM); lanes beyond the 168 real cells recompute cell 0 and are not stored
    int brow[NF];
#pragma unroll
    for (int j = 0; j < NF; j++) {
        int col = wave * 48 + j * 16 + (lane & 15);
        if (col >= NS * 42) col = 0;
        const int s = col / 42, p = col % 42;
        brow[j] = cell_row(s, p / 7, p % 7) * RS + 8 * (lane >> 4);
    }
    f32x4 acc[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; f++)
#pragma unroll
        for (int j = 0; j < NF; j++) acc[f][j] = (f32x4){0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    // operands of slice sl + 1 are fetched while the 72 MFMAs of slice sl issue (two register sets, loop unrolled by 2)
    f16x8 ah[2][MF], al[2][MF], bh[2][NF], bl[2][NF];
    auto fetch = [&](int sl, int b) {
        const int tap = sl >> 2, cs = sl & 3;
        const int toff = ((tap / 3 - 1) * 9 + (tap % 3 - 1)) * RS + cs * 32;
#pragma unroll
        for (int f = 0; f < MF; f++) {
            ah[b][f] = wpk[((size_t)(sl * MF + f) * 2 + 0) * 64 + lane];
            al[b][f] = wpk[((size_t)(sl * MF + f) * 2 + 1) * 64 + lane];
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
    const unsigned long long t1 = __builtin_readcyclecounter();
    float* yb = y + (size_t)blockIdx.x * NS * 42 * CO;
#pragma unroll
    for (int j = 0; j < NF; j++) {
        const int col = wave * 48 + j * 16 + (lane & 15);
        if (col < NS * 42) {
#pragma unroll
            for (int f = 0; f < MF; f++)
#pragma unroll
                for (int r = 0; r < 4; r++) yb[col * CO + 16 * f + 4 * (lane >> 4) + r] = acc[f][j][r] * inv_scale;
        }
    }
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));              \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main() {
    const int WG = 256;
    srand(1);
    auto rnd = []() { return (rand() / (double)RAND_MAX) * 2.0 - 1.0; };
    std::vector<float> w((size_t)CO * CI * 9), x((size_t)WG * NS * CI * 42);
    for (auto& v : w) v = (float)(0.08 * rnd() * rnd());
    for (auto& v : x) v = (float)(1.5 * rnd());
    float wmax = 0;
    for (auto v : w) wmax = fmaxf(wmax, fabsf(v));
    const float scale = exp2f(floorf(log2f(16384.0f / wmax)));
    // pack: slice = tap * 4 + cs; A fragment f: lane -> co = 16 f + lane % 16, ci = 32 cs + 8 (lane / 16) + 0..7
    std::vector<_Float16> pk((size_t)NSL * MF * 2 * 64 * 8);
    for (int sl = 0; sl < NSL; sl++)
        for (int f = 0; f < MF; f++)
            for (int l = 0; l < 64; l++)
                for (int i = 0; i < 8; i++) {
                    const int tap = sl >> 2, cs = sl & 3, co = 16 * f + l % 16, ci = 32 * cs + 8 * (l / 16) + i;
                    const float v = w[((size_t)co * CI + ci) * 9 + tap] * scale;
                    const _Float16 h = (_Float16)v;
                    pk[((((size_t)sl * MF + f) * 2 + 0) * 64 + l) * 8 + i] = h;
                    pk[((((size_t)sl * MF + f) * 2 + 1) * 64 + l) * 8 + i] = (_Float16)(v - (float)h);
                }
    f16x8* dw;
    float *dx, *dy;
    unsigned long long* dc;
    CK(hipMalloc(&dw, pk.size() * 2));
    CK(hipMalloc(&dx, x.size() * 4));
    CK(hipMalloc(&dy, (size_t)WG * NS * 42 * CO * 4));
    CK(hipMalloc(&dc, 8));
    CK(hipMemcpy(dw, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    const size_t smem = (size_t)2 * ROWS * RS * 2;
    CK(hipFuncSetAttribute((const void*)k_conv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        k_conv<<<WG, 256, smem>>>(dw, dx, dy, 1.0f / scale, dc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipGetLastError());
    unsigned long long c = 0;
    CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    std::vector<float> yh((size_t)NS * 42 * CO);
    CK(hipMemcpy(yh.data(), dy, yh.size() * 4, hipMemcpyDeviceToHost));
    // float64 reference for workgroup 0
    double emax = 0, ymax = 0, e32max = 0;
    for (int s = 0; s < NS; s++)
        for (int r = 0; r < 6; r++)
            for (int cc = 0; cc < 7; cc++)
                for (int co = 0; co < CO; co++) {
                    double a = 0;
                    float a32 = 0;
                    for (int ci = 0; ci < CI; ci++)
                        for (int t = 0; t < 9; t++) {
                            const int rr = r + t / 3 - 1, c2 = cc + t % 3 - 1;
                            if (rr < 0 || rr >= 6 || c2 < 0 || c2 >= 7) continue;
                            const float xv = x[((size_t)s * CI + ci) * 42 + rr * 7 + c2], wv = w[((size_t)co * CI + ci) * 9 + t];
                            a += (double)xv * wv;
                            a32 += xv * wv;
                        }
                    const double g = yh[(size_t)(s * 42 + r * 7 + cc) * CO + co];
                    emax = fmax(emax, fabs(g - a));
                    e32max = fmax(e32max, fabs((double)a32 - a));
                    ymax = fmax(ymax, fabs(a));
                }
    const double flop = 2.0 * CO * CI * 9 * 42 * NS;
    printf("LDS %zu B, weight scale 2^%d\n", smem, (int)log2f(scale));
    printf("main loop %llu cycles per workgroup of %d positions (%d MFMAs per wave: %.1f cycles each)\n", c, NS, NSL * MF * NF * 3,
           (double)c / (NSL * MF * NF * 3));
    printf("kernel %.3f ms for %d workgroups: %.1f TFLOP/s algorithmic fp32 (%.1f of executed fp16 MFMA)\n", ms, WG,
           flop * WG / ms / 1e9, 3 * flop * WG / ms / 1e9);
    printf("max |y| %.3f; max abs error vs float64: split-f16 %.3e, sequential fp32 sum %.3e\n", ymax, emax, e32max);
    return emax < 1e-4 * ymax ? 0 : 2;
}
