// Micro-benchmark: what one instruction of each kind costs when it sits between fp32 MFMAs (v_mfma_f32_16x16x4_f32,
// AGPR accumulators, ONE wave per SIMD), gfx950.  Each mode runs 16 MFMAs per trip with its fillers spread one per gap.
//   build: hipcc --offload-arch=gfx950 -O3 -o dbgbuild/mfma_fill tools/ubench/mfma_fill.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(p) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p]) : "v"(a), "v"(b))
#define SB __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float4* gw, unsigned long long* cyc, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-3f;
    __syncthreads();
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f32x4 acc[16];
    float d[16], e[16];
    float4 w[4];
    for (int p = 0; p < 16; p++) acc[p] = (f32x4){0, 0, 0, 0}, d[p] = p, e[p] = 0;
    for (int p = 0; p < 4; p++) w[p] = make_float4(0, 0, 0, 0);
    const float* lp = lds + (threadIdx.x & 63);
    const float4* gp = gw + threadIdx.x;
    asm volatile("s_nop 3");
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int p = 0; p < 16; p++) {
            if constexpr (MODE == 12 || MODE == 13) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p]) : "v"(d[p]), "v"(b));
                if (MODE == 12 || (p & 1) == 0) {
                    f2 x = {e[p & 14], e[(p & 14) + 1]}, y = {b, a};
                    f2 r = x - y;   // v_pk_add_f32 with neg
                    asm volatile("" : "+v"(r));
                    d[(p + 8) & 14] = r[0];
                    d[((p + 8) & 14) + 1] = r[1];
                }
            } else
            if constexpr (MODE == 10 || MODE == 11) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p]) : "v"(d[p]), "v"(b));
                if constexpr (MODE == 10) d[p] = e[p] + b;                  // WAR: overwrites the A operand of the MFMA just issued
                if constexpr (MODE == 11) d[(p + 8) & 15] = e[p] + b;       // overwrites an operand used 8 MFMAs away
            } else
            MFMA(p);
            if constexpr (MODE == 1) d[p] = d[p] + b;                                  // 1 VALU
            if constexpr (MODE == 2) { d[p] = d[p] + b; e[p] = e[p] - a; }             // 2 VALU
            if constexpr (MODE == 3) { d[p] = d[p] + b; e[p] = e[p] - a; d[(p + 8) & 15] = d[(p + 8) & 15] * b; }  // 3 VALU
            if constexpr (MODE == 4) d[p] = lp[p * 64];                                // 1 ds_read_b32 (imm offsets), never read in loop
            if constexpr (MODE == 5) { e[p] = e[p] + d[p]; d[p] = lp[p * 64]; }        // 1 ds_read_b32 + 1 VALU using last trip's read
            if constexpr (MODE == 6) if ((p & 3) == 0) w[p >> 2] = gp[(p >> 2) * 256]; // 1 global dwordx4 per 4 MFMAs (imm offsets)
            if constexpr (MODE == 7) if ((p & 3) == 0) {                               // same + 4 VALU using last trip's load
                e[p] += w[p >> 2].x; e[p + 1] += w[p >> 2].y; e[p + 2] += w[p >> 2].z; e[p + 3] += w[p >> 2].w;
                w[p >> 2] = gp[(p >> 2) * 256];
            }
            if constexpr (MODE == 8) if ((p & 1) == 0) d[p] = lp[p * 64];              // 1 ds_read_b32 per 2 MFMAs
            if constexpr (MODE == 9) if ((p & 7) == 0) {                               // 8 VALU bunched per 8 MFMAs
#pragma unroll
                for (int q = 0; q < 8; q++) d[p + q] = d[p + q] + b;
            }
            SB;
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[15]));
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int p = 0; p < 16; p++) s += acc[p][0] + acc[p][3] + d[p] + e[p];
    for (int p = 0; p < 4; p++) s += w[p].x + w[p].w;
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float* out;
    float4* gw;
    unsigned long long* cyc;
    (void)hipMalloc(&out, 1024 * 256 * 4);
    (void)hipMalloc(&gw, 1 << 20);
    (void)hipMemset(gw, 0, 1 << 20);
    (void)hipMalloc(&cyc, 8);
    const int iters = 2000;
    const char* names[] = {"bare", "1 VALU", "2 VALU", "3 VALU", "1 ds_read_b32", "1 ds_read_b32 + 1 VALU(use)", "1/4 global x4",
                           "1/4 global x4 + 1 VALU(use)", "1/2 ds_read_b32", "8 VALU bunched / 8 MFMA", "1 VALU WAR on MFMA srcA", "1 VALU write srcA of MFMA+8", "1 v_pk_add_f32", "1/2 v_pk_add_f32"};
    for (int m = 0; m < 14; m++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (m) {
#define C(M) case M: k<M><<<256, 256>>>(out, gw, cyc, iters); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13)
            }
            (void)hipDeviceSynchronize();
        }
        unsigned long long c;
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-30s %6.1f cyc/MFMA\n", names[m], c / (16.0 * iters));
    }
    return 0;
}
