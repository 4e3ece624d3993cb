// Issue cost of the epilogue's VALU instructions on gfx950, one wave per SIMD, 8 independent chains, 32 instructions per loop trip, no MFMA in flight:
// shader cycles per instruction for v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_exp_f32, v_rcp_f32, v_cvt_pk_f16_f32,
// v_fma_mix_f32, v_max3_f32.  (Round 2, K3: the split kernel's layer epilogue is VALU-issue bound; this says which
// instructions to spend.)
// build: hipcc --offload-arch=gfx950 -O3 -o dbgbuild/valu_rate tools/ubench/valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CHAIN8(INS)                                                                                               \
    for (int i = 0; i < n; i++) {                                                                                 \
        asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                     INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                     : "v"(c0), "v"(c1));                                                                         \
    }
#define CHAINP(INS)                                                                                               \
    for (int i = 0; i < n; i++) {                                                                                 \
        asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                     INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                     : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) \
                     : "v"(d0), "v"(d1));                                                                         \
    }
#define S(x) #x
#define FMA(p) "v_fma_f32 %" S(p) ", %" S(p) ", %8, %9\n"
#define EXP(p) "v_exp_f32 %" S(p) ", %" S(p) "\n"
#define RCP(p) "v_rcp_f32 %" S(p) ", %" S(p) "\n"
#define MAX3(p) "v_max3_f32 %" S(p) ", |%" S(p) "|, |%8|, |%9|\n"
#define MIX(p) "v_fma_mix_f32 %" S(p) ", %8, %9, %" S(p) " op_sel_hi:[1,0,0]\n"
#define CVT(p) "v_cvt_pk_f16_f32 %" S(p) ", %" S(p) ", %8\n"
#define PKFMA(p) "v_pk_fma_f32 %" S(p) ", %" S(p) ", %8, %9\n"
#define PKMUL(p) "v_pk_mul_f32 %" S(p) ", %" S(p) ", %8\n"
#define PKADD(p) "v_pk_add_f32 %" S(p) ", %" S(p) ", %8\n"

__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int n, int kind) {
    float a[8];
    f32x2 b[8];
    for (int p = 0; p < 8; p++) {
        a[p] = threadIdx.x * 1e-3f + p;
        b[p] = (f32x2){a[p], a[p] + 0.5f};
    }
    const float c0 = 1.0001f, c1 = 0.25f;
    const f32x2 d0 = {1.0001f, 0.9999f}, d1 = {0.25f, -0.25f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    switch (kind) {
    case 0: CHAIN8(FMA) break;
    case 1: CHAIN8(EXP) break;
    case 2: CHAIN8(RCP) break;
    case 3: CHAIN8(MAX3) break;
    case 4: CHAIN8(MIX) break;
    case 5: CHAIN8(CVT) break;
    case 6: CHAINP(PKFMA) break;
    case 7: CHAINP(PKMUL) break;
    default: CHAINP(PKADD) break;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int p = 0; p < 8; p++) s += a[p] + b[p][0] + b[p][1];
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    float* out;
    unsigned long long *cyc, h = 0;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&cyc, 64);
    const char* names[9] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_max3_f32 (abs)", "v_fma_mix_f32", "v_cvt_pk_f16_f32",
                            "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"};
    const int n = 4096;
    for (int kind = 0; kind < 9; kind++) {
        for (int rep = 0; rep < 2; rep++) {
            k<<<256, 256>>>(out, cyc, n, kind);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-20s %.2f cycles per instruction (one wave per SIMD, 8 independent chains)\n", names[kind], (double)h / (n * 32.0));
    }
    return 0;
}
