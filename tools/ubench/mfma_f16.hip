// Micro-benchmark for the round-2 K3 candidate: issue cost of the fp16 MFMA forms on gfx950, ONE
// wave per SIMD, back to back, and what a VALU instruction between them costs (fp32 MFMA and VALU do not overlap in one
// wave; do fp16 MFMA and VALU?).
//   build: hipcc --offload-arch=gfx950 -O3 -o dbgbuild/mfma_f16 tools/ubench/mfma_f16.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    const _Float16 ha = (_Float16)(threadIdx.x * 1e-3f), hb = (_Float16)(1.0f + threadIdx.x * 1e-4f);
    f16x4 a4 = {ha, hb, ha, hb}, b4 = {hb, ha, hb, ha};
    f16x8 a8 = {ha, hb, ha, hb, ha, hb, ha, hb}, b8 = {hb, ha, hb, ha, hb, ha, hb, ha};
    float fb = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = 0, t1 = 0;
    float s = 0;
    if constexpr (MODE == 0) {  // 16x16x16 f16 (8 192 flop), 16 accumulators
        f32x4 acc[16];
        for (int p = 0; p < 16; p++) acc[p] = (f32x4){0, 0, 0, 0};
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int p = 0; p < 16; p++) acc[p] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[p], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        for (int p = 0; p < 16; p++) s += acc[p][0] + acc[p][3];
    } else if constexpr (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) {  // 16x16x32 f16 (16 384 flop) [+ VALU / LDS]
        __shared__ float lds[4096];
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
        __syncthreads();
        f32x4 acc[16];
        float d[16];
        for (int p = 0; p < 16; p++) acc[p] = (f32x4){0, 0, 0, 0}, d[p] = p;
        int off = threadIdx.x;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int p = 0; p < 16; p++) {
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[p], 0, 0, 0);
                if constexpr (MODE == 3) d[p] = d[p] + fb;
                if constexpr (MODE == 4) {
                    d[p] = d[p] + fb;
                    d[(p + 5) & 15] = d[(p + 5) & 15] - fb;
                }
                if constexpr (MODE == 5) d[p] = lds[(off + p * 67) & 4095];
                if constexpr (MODE != 1) __builtin_amdgcn_sched_barrier(0);
            }
            off += 16;
        }
        t1 = __builtin_readcyclecounter();
        for (int p = 0; p < 16; p++) s += acc[p][0] + acc[p][3] + d[p];
    } else if constexpr (MODE == 2) {  // 32x32x16 f16 (32 768 flop), 4 accumulators
        f32x16 acc[4];
        for (int p = 0; p < 4; p++)
            for (int r = 0; r < 16; r++) acc[p][r] = 0;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int p = 0; p < 4; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[p], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        for (int p = 0; p < 4; p++) s += acc[p][0] + acc[p][15];
    }
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float* out;
    unsigned long long* cyc;
    if (hipMalloc(&out, 1024 * 256 * 4) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 1;
    const int iters = 2000;
    const char* names[] = {"16x16x16 f16 x16 acc", "16x16x32 f16 x16 acc", "32x32x16 f16 x4 acc", "16x16x32 f16 + 1 VALU",
                           "16x16x32 f16 + 2 VALU", "16x16x32 f16 + 1 ds_read"};
    const double flop[] = {8192, 16384, 32768, 16384, 16384, 16384};
    for (int grid : {1, 256}) {
        for (int m = 0; m < 6; m++) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (m == 0) k<0><<<grid, 256>>>(out, cyc, iters);
                if (m == 1) k<1><<<grid, 256>>>(out, cyc, iters);
                if (m == 2) k<2><<<grid, 256>>>(out, cyc, iters);
                if (m == 3) k<3><<<grid, 256>>>(out, cyc, iters);
                if (m == 4) k<4><<<grid, 256>>>(out, cyc, iters);
                if (m == 5) k<5><<<grid, 256>>>(out, cyc, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            unsigned long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double nm = (m == 2 ? 4.0 : 16.0) * iters;
            printf("grid %4d  %-26s  %6.1f cyc/MFMA   %8.3f ms  %7.1f TFLOP/s\n", grid, names[m], c / nm, ms,
                   flop[m] * nm * 4 * grid / ms / 1e9);
        }
    }
    return 0;
}
