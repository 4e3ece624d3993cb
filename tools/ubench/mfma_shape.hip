// v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 on the SAME wave tile (64 x 64 outputs, 64 accumulator registers per
// lane) and the SAME data, every SIMD of the chip issuing back to back -- the question behind K3's tile loops (VERDICT round 2,
// item 3; MI355X_MICROARCH.md "DVFS give-back" (7): the clock the chip holds under load can depend on the MFMA shape, and zero
// or trivial operands miss this).  Round 2's tools/ubench/mfma_f16.hip multiplied ONE operand pair into sixteen accumulators
// inside a rolled loop and read 27 cycles per 16x16x32 where the documented issue cost is 16: not a property of the
// instruction (this benchmark reads 16.x on the same chip); its conclusion is withdrawn.
//   per 32-deep k-slice of the tile:  32x32x16: 2 k-steps x (2 x 2 fragments)       =  8 MFMAs x 32 768 flop
//                                     16x16x32: 1 k-step  x (4 x 4 fragments)       = 16 MFMAs x 16 384 flop
//   operands per lane and k-slice: 4 A + 4 B registers quads (f16x8) in both shapes; NSET independent sets are cycled so that
//   consecutive k-slices multiply different data (per-lane distinct values, kinds as in mfma_power.hip).
// Prints launch time, shader cycles per MFMA and per k-slice, the clock they imply and the executed TFLOP/s chip-wide.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_ab/mfma_shape tools/ubench/mfma_shape.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned long long mix(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

constexpr int NSET = 4;

// kind 0 zeros, 1 one constant pair, 2 random mantissas with one exponent, 3 K3-like spread (weight-like hi halves around
// 2^13 with either sign, activation-like values over 2^0 .. 2^13)
__device__ inline void fill(f16x8 (&a)[NSET][4], f16x8 (&b)[NSET][4], int kind) {
    for (int s = 0; s < NSET; s++)
        for (int p = 0; p < 4; p++)
            for (int i = 0; i < 8; i++) {
                const unsigned long long h = mix(((unsigned long long)(blockIdx.x * 256 + threadIdx.x) << 10) + (s * 4 + p) * 8 + i + 1);
                float va = 0.f, vb = 0.f;
                if (kind == 1) va = 1.25f, vb = 0.75f;
                if (kind == 2) va = 1.0f + (float)(h & 1023) / 1024.0f, vb = 1.0f + (float)((h >> 20) & 1023) / 1024.0f;
                if (kind >= 3) {
                    va = (8192.0f + (float)(h & 8191)) * ((h >> 40) & 1 ? -1.f : 1.f) * __builtin_exp2f(-(float)((h >> 44) & 3));
                    vb = (1.0f + (float)((h >> 20) & 1023) / 1024.0f) * __builtin_exp2f((float)((h >> 50) % 14)) * ((h >> 41) & 1 ? -1.f : 1.f);
                }
                a[s][p][i] = (_Float16)va;
                b[s][p][i] = (_Float16)vb;
            }
}

// SHAPE 32: a[s][2 kh + f], b[s][2 kh + j]; SHAPE 16: a[s][f], b[s][j]; SHAPE 17: as 16 with the operand bits read as bf16
// (v_mfma_f32_16x16x32_bf16, the form MI355X_MICROARCH.md quotes 16 cycles for); SHAPE 18: as 16, j outer / f inner
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int n, int kind) {
    f16x8 a[NSET][4], b[NSET][4];
    fill(a, b, kind);
    float s = 0;
    unsigned long long t0, t1;
    if constexpr (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int f = 0; f < 2; f++)
            for (int j = 0; j < 2; j++)
                for (int r = 0; r < 16; r++) acc[f][j][r] = 0;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int st = 0; st < NSET; st++)
#pragma unroll
                for (int kh = 0; kh < 2; kh++)
#pragma unroll
                    for (int f = 0; f < 2; f++)
#pragma unroll
                        for (int j = 0; j < 2; j++)
                            acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[st][2 * kh + f], b[st][2 * kh + j], acc[f][j], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        for (int f = 0; f < 2; f++)
            for (int j = 0; j < 2; j++) s += acc[f][j][0] + acc[f][j][9];
    } else {
        f32x4 acc[4][4];
        for (int f = 0; f < 4; f++)
            for (int j = 0; j < 4; j++) acc[f][j] = (f32x4){0, 0, 0, 0};
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int st = 0; st < NSET; st++)
#pragma unroll
                for (int f = 0; f < 4; f++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        if constexpr (SHAPE == 16) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[st][f], b[st][j], acc[f][j], 0, 0, 0);
                        if constexpr (SHAPE == 17)
                            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[st][f]), __builtin_bit_cast(bf16x8, b[st][j]), acc[f][j], 0, 0, 0);
                        if constexpr (SHAPE == 18) acc[j][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[st][j], b[st][f], acc[j][f], 0, 0, 0);
                    }
        }
        t1 = __builtin_readcyclecounter();
        for (int f = 0; f < 4; f++)
            for (int j = 0; j < 4; j++) s += acc[f][j][0] + acc[f][j][3];
    }
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
    const char* names[4] = {"zeros", "one constant pair", "random mantissas, one exponent", "K3-like values"};
    const int n = 8192;   // x NSET k-slices per trip: ~5-8 ms per launch, long enough for the clock to settle
    for (int rep_all = 0; rep_all < 2; rep_all++)
        for (int kind = 0; kind < 4; kind++)
            for (int shape : {32, 16, 17, 18}) {
                float ms = 0;
                for (int rep = 0; rep < 3; rep++) {
                    (void)hipEventRecord(e0);
                    if (shape == 32) k<32><<<256, 256>>>(out, cyc, n, kind);
                    else if (shape == 16) k<16><<<256, 256>>>(out, cyc, n, kind);
                    else if (shape == 17) k<17><<<256, 256>>>(out, cyc, n, kind);
                    else k<18><<<256, 256>>>(out, cyc, n, kind);
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    (void)hipEventElapsedTime(&ms, e0, e1);
                }
                (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
                const double slices = (double)n * NSET, mf = slices * (shape == 32 ? 8 : 16), flop = slices * 262144.0 * 1024.0;
                printf("%-32s %2d(%s)  %8.1f us  %5.1f cycles per MFMA  %6.1f per k-slice  clock %.2f GHz  %5.0f TFLOP/s executed chip-wide\n",
                       names[kind], shape == 32 ? 32 : 16, shape == 17 ? "bf16" : (shape == 18 ? "f16 j-outer" : "f16"), ms * 1e3, (double)h / mf, (double)h / slices, (double)h / (ms * 1e6),
                       flop / (ms * 1e-3) / 1e12);
            }
    return 0;
}
