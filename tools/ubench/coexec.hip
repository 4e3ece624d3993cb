// Do one wave's MFMAs and ANOTHER wave's VALU / transcendental work overlap on one SIMD of gfx950?  (Round 2, K3: the split
// kernel's time is MFMA + epilogue arithmetic + waits added up; running a second workgroup per CU under the first one's MFMAs
// only pays if this is true.)  512-thread workgroup = two waves per SIMD; waves 0-3 run NM x 8 MFMAs (32x32x16 f16, 8
// independent accumulators), waves 4-7 run NV x 8 blocks of VALU work.  kind 0: {v_exp, v_rcp, 3 mul/fma} (SiLU-like),
// kind 1: 6 plain fma.  Modes: 1 MFMA waves only, 2 VALU waves only, 3 both.  Times in shader cycles (s_memtime) of wave 0 / 4
// and in microseconds for the launch, one workgroup per CU on every CU.  prio 0: equal priority; prio 1: the arithmetic waves
// run at s_setprio 3; prio 2 / 3 / 4: the MFMA wave idles on the scalar side (s_nop / s_sleep) after every MFMA.
// Result (MI355X): see profiles/r02_k3_split_breakdown.txt and DESIGN.md 4.2.
// build: hipcc --offload-arch=gfx950 -O3 -o dbgbuild/coexec tools/ubench/coexec.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int prio>   // a template parameter: a run-time test between two MFMAs would itself open issue gaps
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int nm, int nv, int mode, int kind) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (mode & 1) {
            const _Float16 ha = (_Float16)(threadIdx.x * 1e-3f), hb = (_Float16)(1.0f + threadIdx.x * 1e-4f);
            f16x8 a8 = {ha, hb, ha, hb, ha, hb, ha, hb}, b8 = {hb, ha, hb, ha, hb, ha, hb, ha};
            f32x16 acc[8];
            for (int p = 0; p < 8; p++)
                for (int r = 0; r < 16; r++) acc[p][r] = 0;
            for (int i = 0; i < nm; i++) {
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[p], 0, 0, 0);
                    // prio 2 / 3: the MFMA wave idles on the scalar side while its MFMA executes, so that its NEXT MFMA does
                    // not sit at the head of the vector issue port waiting for the matrix pipe
                    if (prio == 2) { asm volatile("s_nop 15"); asm volatile("s_nop 3"); }
                    if (prio == 3) { asm volatile("s_nop 15"); asm volatile("s_nop 7"); }
                    if (prio == 4) { __builtin_amdgcn_s_sleep(0); }
                }
            }
            for (int p = 0; p < 8; p++) s += acc[p][0] + acc[p][7];
        }
    } else if (mode & 2) {
        if (prio == 1) __builtin_amdgcn_s_setprio(3);   // the arithmetic wave outranks the MFMA wave at the issue arbiter
        float v[8];
        for (int p = 0; p < 8; p++) v[p] = threadIdx.x * 1e-3f + p;
        if (kind == 0) {
            for (int i = 0; i < nv; i++) {
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    float e = __expf(-v[p]);
                    float r = __builtin_amdgcn_rcpf(1.0f + e);
                    v[p] = v[p] * r * 0.999f + 0.25f;
                }
            }
        } else {
            for (int i = 0; i < nv; i++) {
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    float x = v[p];
                    x = x * 1.0001f + 0.25f;
                    x = x * 0.9999f - 0.25f;
                    x = x * 1.0001f + 0.125f;
                    x = x * 0.9999f - 0.125f;
                    x = x * 1.0001f + 0.5f;
                    x = x * 0.9999f - 0.5f;
                    v[p] = x;
                }
            }
        }
        for (int p = 0; p < 8; p++) s += v[p];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    float* out;
    unsigned long long *cyc, h[8];
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&cyc, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int nm = 2048;
    for (int prio = 0; prio < 5; prio++)
    for (int kind = 0; kind < 2; kind++)
        for (int nv = 1024; nv <= 4096; nv *= 2)
            for (int mode = 1; mode <= 3; mode++) {
                if (prio >= 2 && (kind != 0 || nv != 2048)) continue;   // the idling variants: one line each
                for (int rep = 0; rep < 2; rep++) {
                    (void)hipEventRecord(e0);
                    switch (prio) {
                    case 0: k<0><<<256, 512>>>(out, cyc, nm, nv, mode, kind); break;
                    case 1: k<1><<<256, 512>>>(out, cyc, nm, nv, mode, kind); break;
                    case 2: k<2><<<256, 512>>>(out, cyc, nm, nv, mode, kind); break;
                    case 3: k<3><<<256, 512>>>(out, cyc, nm, nv, mode, kind); break;
                    default: k<4><<<256, 512>>>(out, cyc, nm, nv, mode, kind); break;
                    }
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                }
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
                printf("prio %d kind %d nm %d x8 MFMAs, nv %d x8 blocks, mode %d (%s): %.1f us; cycles wave0 %llu (%.1f per MFMA) wave4 %llu\n",
                       prio, kind, nm, nv, mode, mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms * 1e3, h[0],
                       (double)h[0] / (nm * 8.0), h[4]);
            }
    return 0;
}
