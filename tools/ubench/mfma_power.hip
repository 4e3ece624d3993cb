// What the fp16 matrix pipes of an MI355X sustain when EVERY SIMD issues v_mfma_f32_32x32x16_f16 back to back -- as a
// function of the DATA.  (Round 2, K3: operand-removal builds showed that the split kernel's launch time follows the bits
// toggling inside the matrix pipe, not operand traffic and not issue slots; this measures the ceiling that sets.)
// One wave per SIMD on all 256 CUs, 8 accumulators, 8 A and 8 B operand registers per lane; per MFMA the operands are
//   kind 0  zeros                         kind 1  one constant pair for every MFMA
//   kind 2  random fp16 values in [1, 2) (random mantissas, one exponent), a different (A, B) pair for every MFMA
//   kind 3  random fp16 values with K3-like spread (weights-like hi halves around 2^13, activation-like values over 2^0..2^13)
//   kind 4  as 3, but two MFMAs in a row share their A operand (the order the split kernel issues them in)
// Prints the launch time, shader cycles per MFMA, the clock they imply and the executed TFLOP/s chip-wide.
// build: hipcc --offload-arch=gfx950 -O3 -o dbgbuild/mfma_power tools/ubench/mfma_power.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned long long mix(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int n, int kind) {
    f16x8 a[8], b[8];
    for (int p = 0; p < 8; p++)
        for (int i = 0; i < 8; i++) {
            const unsigned long long h = mix(((unsigned long long)(blockIdx.x * 256 + threadIdx.x) << 8) + p * 8 + i + 1);
            float va = 0.f, vb = 0.f;
            if (kind == 1) va = 1.25f, vb = 0.75f;
            if (kind == 2) va = 1.0f + (float)(h & 1023) / 1024.0f, vb = 1.0f + (float)((h >> 20) & 1023) / 1024.0f;
            if (kind >= 3) {
                va = (8192.0f + (float)(h & 8191)) * ((h >> 40) & 1 ? -1.f : 1.f) * __builtin_exp2f(-(float)((h >> 44) & 3));
                vb = (1.0f + (float)((h >> 20) & 1023) / 1024.0f) * __builtin_exp2f((float)((h >> 50) % 14)) * ((h >> 41) & 1 ? -1.f : 1.f);
            }
            a[p][i] = (_Float16)va;
            b[p][i] = (_Float16)vb;
        }
    f32x16 acc[8];
    for (int p = 0; p < 8; p++)
        for (int r = 0; r < 16; r++) acc[p][r] = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (kind == 4) {
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int p = 0; p < 8; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[p >> 1], b[p], acc[p], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < 8; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[4 + (p >> 1)], b[(p + 3) & 7], acc[p], 0, 0, 0);
        }
    } else {
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int p = 0; p < 8; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[p], b[(p * 3) & 7], acc[p], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < 8; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(p + 5) & 7], b[(p + 1) & 7], acc[p], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int p = 0; p < 8; p++) s += acc[p][0] + acc[p][9];
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    float* out;
    unsigned long long *cyc, h = 0;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&cyc, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const char* names[5] = {"zeros", "one constant pair", "random mantissas, one exponent", "K3-like values", "K3-like, A shared by 2 MFMAs in a row"};
    const int n = 16384;   // x 16 MFMAs per wave: ~4-6 ms per launch, long enough for the clock to settle
    for (int kind = 0; kind < 5; kind++) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            k<<<256, 256>>>(out, cyc, n, kind);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double mf = (double)n * 16.0, flop = mf * 32768.0 * 1024.0;
        printf("%-40s %8.1f us  %.1f cycles per MFMA  clock %.2f GHz  %.0f TFLOP/s executed chip-wide\n", names[kind], ms * 1e3,
               (double)h / mf, (double)h / (ms * 1e6), flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
