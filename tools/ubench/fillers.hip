// Same-wave fillers: how many independent VALU instructions hide under a v_mfma_f32_32x32x16_f16 of the SAME wave on gfx950?
// One wave per SIMD (256 threads), 8 accumulators, NF fillers after every MFMA (plain v_fma_f32, or v_exp_f32), pinned with
// sched_barrier so that the compiler keeps the interleave.  Prints shader cycles per MFMA.
// build: hipcc --offload-arch=gfx950 -O3 -o dbgbuild/fillers tools/ubench/fillers.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NF, int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int nm) {
    const _Float16 ha = (_Float16)(threadIdx.x * 1e-3f), hb = (_Float16)(1.0f + threadIdx.x * 1e-4f);
    f16x8 a8 = {ha, hb, ha, hb, ha, hb, ha, hb}, b8 = {hb, ha, hb, ha, hb, ha, hb, ha};
    f32x16 acc[8];
    float v[8];
    for (int p = 0; p < 8; p++) {
        v[p] = threadIdx.x * 1e-3f + p;
        for (int r = 0; r < 16; r++) acc[p][r] = 0;
    }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < nm; i++) {
#pragma unroll
        for (int p = 0; p < 8; p++) {
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[p], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; f++) {
                if (KIND == 0) v[(p + f) & 7] = v[(p + f) & 7] * 1.0001f + 0.25f;
                else v[(p + f) & 7] = __expf(v[(p + f) & 7]) * 0.5f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int p = 0; p < 8; p++) s += acc[p][0] + acc[p][7] + v[p];
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NF, int KIND>
void run(float* out, unsigned long long* cyc) {
    const int nm = 1024;
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; rep++) {
        k<NF, KIND><<<256, 256>>>(out, cyc, nm);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s fillers per MFMA: %d -> %.1f cycles per MFMA\n", KIND == 0 ? "v_fma_f32" : "v_exp_f32+v_mul", NF, (double)h / (nm * 8.0));
}

int main() {
    float* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&cyc, 64);
    run<0, 0>(out, cyc);
    run<1, 0>(out, cyc);
    run<2, 0>(out, cyc);
    run<4, 0>(out, cyc);
    run<6, 0>(out, cyc);
    run<8, 0>(out, cyc);
    run<1, 1>(out, cyc);
    run<2, 1>(out, cyc);
    run<4, 1>(out, cyc);
    return 0;
}
