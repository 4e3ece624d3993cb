// Micro-benchmark: cycles per fp32 MFMA issued back-to-back by ONE wave per SIMD (gfx950), by form.
//   build: hipcc --offload-arch=gfx950 -O3 -o dbgbuild/mfma_rate tools/ubench/mfma_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = 0, t1 = 0;
    if constexpr (MODE == 0) {  // builtin 16x16x4, 16 accumulators, VGPR form
        f32x4 acc[16];
        for (int p = 0; p < 16; p++) acc[p] = (f32x4){0, 0, 0, 0};
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int p = 0; p < 16; p++) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[p], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int p = 0; p < 16; p++) s += acc[p][0] + acc[p][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (MODE == 1) {  // asm, AGPR accumulators
        f32x4 acc[16];
        for (int p = 0; p < 16; p++) acc[p] = (f32x4){0, 0, 0, 0};
        asm volatile("s_nop 3");
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int p = 0; p < 16; p++)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[15]));
        t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int p = 0; p < 16; p++) s += acc[p][0] + acc[p][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (MODE == 2) {  // builtin 32x32x2, 4 accumulators
        f32x16 acc[4];
        for (int p = 0; p < 4; p++)
            for (int r = 0; r < 16; r++) acc[p][r] = 0;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int p = 0; p < 4; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int p = 0; p < 4; p++) s += acc[p][0] + acc[p][15];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (MODE == 3) {  // asm, VGPR accumulators
        f32x4 acc[16];
        for (int p = 0; p < 16; p++) acc[p] = (f32x4){0, 0, 0, 0};
        asm volatile("s_nop 3");
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int p = 0; p < 16; p++)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[p]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[15]));
        t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int p = 0; p < 16; p++) s += acc[p][0] + acc[p][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if constexpr (MODE == 4) {  // builtin 16x16x4 with 2 v_add fillers per MFMA
        f32x4 acc[16];
        float f[8];
        for (int p = 0; p < 16; p++) acc[p] = (f32x4){0, 0, 0, 0};
        for (int p = 0; p < 8; p++) f[p] = a + p;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int p = 0; p < 16; p++) {
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[p], 0, 0, 0);
                f[p & 7] = f[p & 7] + b;
                f[(p + 3) & 7] = f[(p + 3) & 7] - a;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int p = 0; p < 16; p++) s += acc[p][0] + acc[p][3];
        for (int p = 0; p < 8; p++) s += f[p];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
    else if constexpr (MODE == 5 || MODE == 6 || MODE == 7) {  // 5: + 1 ds_read_b32 / MFMA; 6: + 1 global dwordx4 / 4 MFMAs; 7: 1 VALU/MFMA
        __shared__ float lds[4096];
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
        __syncthreads();
        f32x4 acc[16];
        float d[16];
        float4 w[4];
        const float4* gp = reinterpret_cast<const float4*>(out) + threadIdx.x;
        for (int p = 0; p < 16; p++) acc[p] = (f32x4){0, 0, 0, 0}, d[p] = 0;
        for (int p = 0; p < 4; p++) w[p] = make_float4(0, 0, 0, 0);
        int off = threadIdx.x;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int p = 0; p < 16; p++) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p]) : "v"(a), "v"(b));
                if constexpr (MODE == 5) d[p] = lds[(off + p * 67) & 4095];
                if constexpr (MODE == 6) if ((p & 3) == 0) w[p >> 2] = gp[((i & 63) * 4 + (p >> 2)) * 256];
                if constexpr (MODE == 7) d[p] = d[p] + b;
                __builtin_amdgcn_sched_barrier(0);
            }
            off += 16;
        }
        asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[15]));
        t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int p = 0; p < 16; p++) s += acc[p][0] + acc[p][3] + d[p];
        for (int p = 0; p < 4; p++) s += w[p].x + w[p].w;
        if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    const char* names[] = {"builtin 16x16x4 x16 acc", "asm +a 16x16x4 x16 acc", "builtin 32x32x2 x4 acc", "asm +v 16x16x4 x16 acc",
                           "builtin 16x16x4 + 2 VALU", "asm + 1 ds_read_b32", "asm + 1/4 global x4", "asm + 1 VALU"};
    for (int grid : {1, 256}) {
        for (int m = 0; m < 8; m++) {
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
                if (m == 6) k<6><<<grid, 256>>>(out, cyc, iters);
                if (m == 7) k<7><<<grid, 256>>>(out, cyc, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            unsigned long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double nm = (m == 2 ? 4.0 : 16.0) * iters;
            const double flops = (m == 2 ? 4096.0 : 2048.0) * nm * 4 * grid;
            printf("grid %4d  %-26s  %6.1f cyc/MFMA   %8.3f ms  %7.1f TFLOP/s\n", grid, names[m], c / nm, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
