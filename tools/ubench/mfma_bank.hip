// Micro-benchmark: does the cost of a VALU op between fp32 MFMAs depend on WHICH registers it touches?  (gfx950)
// Hand-written body with explicit registers: 16 x { v_mfma a[4p..], vA, vB ; v_add_f32 vD, vS0, vS1 }.
//   build: hipcc --offload-arch=gfx950 -O3 -o dbgbuild/mfma_bank tools/ubench/mfma_bank.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>

#define STR2(x) #x
#define STR(x) STR2(x)
// slice p: MFMA on accumulator tile p with A = v[A], B = v[B]; VALU: v[D] = v[S0] + v[S1]
#define SL(p, A, B, D, S0, S1)                                                                                   \
    "v_mfma_f32_16x16x4_f32 a[" STR(p * 4) ":" STR(p * 4 + 3) "], v" STR(A) ", v" STR(B) ", a[" STR(p * 4) ":" STR( \
        p * 4 + 3) "]\n\t"                                                                                       \
                   "v_add_f32 v" STR(D) ", v" STR(S0) ", v" STR(S1) "\n\t"
#define SLN(p, A, B) \
    "v_mfma_f32_16x16x4_f32 a[" STR(p * 4) ":" STR(p * 4 + 3) "], v" STR(A) ", v" STR(B) ", a[" STR(p * 4) ":" STR(p * 4 + 3) "]\n\t"

#define BODY16(A, B, D, S0, S1)                                                                                    \
    SL(0, A, B, D, S0, S1) SL(1, A, B, D, S0, S1) SL(2, A, B, D, S0, S1) SL(3, A, B, D, S0, S1) SL(4, A, B, D, S0, S1)  \
        SL(5, A, B, D, S0, S1) SL(6, A, B, D, S0, S1) SL(7, A, B, D, S0, S1) SL(8, A, B, D, S0, S1)                   \
            SL(9, A, B, D, S0, S1) SL(10, A, B, D, S0, S1) SL(11, A, B, D, S0, S1) SL(12, A, B, D, S0, S1)            \
                SL(13, A, B, D, S0, S1) SL(14, A, B, D, S0, S1) SL(15, A, B, D, S0, S1)

#define VA(D, S0, S1) "v_add_f32 v" STR(D) ", v" STR(S0) ", v" STR(S1) "\n\t"
#define VA4 VA(50, 51, 53) VA(46, 47, 49) VA(54, 55, 53) VA(42, 43, 45)
#define M4(p) SLN(p, 40, 41) SLN(p + 1, 40, 41) SLN(p + 2, 40, 41) SLN(p + 3, 40, 41)
#define DS(D, A) "ds_read_b32 v" STR(D) ", v" STR(A) "\n\t"
#define CLOB                                                                                                         \
    "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "a0", \
        "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", \
        "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34",     \
        "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50",     \
        "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* cyc, int iters) {
    // v40.. initialised to small finite values
    asm volatile(
        "v_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 0.5\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\tv_mov_b32 v44, 0\n\t"
        "v_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\tv_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\t"
        "v_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\t" ::
            : CLOB);
    __shared__ float lds[256];
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    asm volatile("v_and_b32 v56, 63, %0\n\tv_lshlrev_b32 v56, 2, v56" ::"v"(threadIdx.x) : "v56");
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if constexpr (MODE == 0) asm volatile(BODY16(40, 41, 50, 51, 53)::: CLOB);   // all banks distinct: A0 B1 D2 S3 S1
        if constexpr (MODE == 1) asm volatile(BODY16(40, 41, 50, 41, 51)::: CLOB);   // VALU reads the MFMA's B register
        if constexpr (MODE == 2) asm volatile(BODY16(40, 41, 50, 44, 45)::: CLOB);   // VALU sources in the banks of A and B
        if constexpr (MODE == 3) asm volatile(BODY16(40, 41, 44, 46, 47)::: CLOB);   // VALU dst in bank of A
        if constexpr (MODE == 4) asm volatile(BODY16(40, 41, 50, 50, 51)::: CLOB);   // VALU accumulates into itself (chain)
        if constexpr (MODE == 5) asm volatile(BODY16(40, 50, 50, 46, 47)::: CLOB);   // VALU writes the register the MFMAs read as B
        if constexpr (MODE == 6)                                                      // bare
            asm volatile(SLN(0, 40, 41) SLN(1, 40, 41) SLN(2, 40, 41) SLN(3, 40, 41) SLN(4, 40, 41) SLN(5, 40, 41) SLN(6, 40, 41)
                             SLN(7, 40, 41) SLN(8, 40, 41) SLN(9, 40, 41) SLN(10, 40, 41) SLN(11, 40, 41) SLN(12, 40, 41)
                                 SLN(13, 40, 41) SLN(14, 40, 41) SLN(15, 40, 41)::
                             : CLOB);
        if constexpr (MODE == 7)   // 16 MFMAs, then 16 VALU
            asm volatile(M4(0) M4(4) M4(8) M4(12) VA4 VA4 VA4 VA4 ::: CLOB);
        if constexpr (MODE == 8)   // 4 x (4 MFMAs, 4 VALU)
            asm volatile(M4(0) VA4 M4(4) VA4 M4(8) VA4 M4(12) VA4 ::: CLOB);
        if constexpr (MODE == 9)   // 16 x (MFMA, 2 VALU)
            asm volatile(SLN(0, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(1, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(2, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(3, 40, 41) VA(50, 51, 53) VA(46, 47, 49)
                         SLN(4, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(5, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(6, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(7, 40, 41) VA(50, 51, 53) VA(46, 47, 49)
                         SLN(8, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(9, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(10, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(11, 40, 41) VA(50, 51, 53) VA(46, 47, 49)
                         SLN(12, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(13, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(14, 40, 41) VA(50, 51, 53) VA(46, 47, 49) SLN(15, 40, 41) VA(50, 51, 53) VA(46, 47, 49) ::: CLOB);
        if constexpr (MODE == 10)  // 16 MFMAs with one VALU after every 4th
            asm volatile(M4(0) VA(50, 51, 53) M4(4) VA(46, 47, 49) M4(8) VA(54, 55, 53) M4(12) VA(42, 43, 45) ::: CLOB);
        if constexpr (MODE == 11)  // 16 x (MFMA, s_nop 0)
            asm volatile(SLN(0, 40, 41) "s_nop 0\n\t" SLN(1, 40, 41) "s_nop 0\n\t" SLN(2, 40, 41) "s_nop 0\n\t" SLN(3, 40, 41) "s_nop 0\n\t" SLN(4, 40, 41) "s_nop 0\n\t" SLN(5, 40, 41) "s_nop 0\n\t" SLN(6, 40, 41) "s_nop 0\n\t" SLN(7, 40, 41) "s_nop 0\n\t"
                         SLN(8, 40, 41) "s_nop 0\n\t" SLN(9, 40, 41) "s_nop 0\n\t" SLN(10, 40, 41) "s_nop 0\n\t" SLN(11, 40, 41) "s_nop 0\n\t" SLN(12, 40, 41) "s_nop 0\n\t" SLN(13, 40, 41) "s_nop 0\n\t" SLN(14, 40, 41) "s_nop 0\n\t" SLN(15, 40, 41) "s_nop 0\n\t" ::: CLOB);
        if constexpr (MODE == 12)  // 16 x (MFMA, ds_read_b32) -- v56 = address 0..252
            asm volatile(SLN(0, 40, 41) DS(50, 56) SLN(1, 40, 41) DS(51, 56) SLN(2, 40, 41) DS(52, 56) SLN(3, 40, 41) DS(53, 56) SLN(4, 40, 41) DS(50, 56) SLN(5, 40, 41) DS(51, 56) SLN(6, 40, 41) DS(52, 56) SLN(7, 40, 41) DS(53, 56)
                         SLN(8, 40, 41) DS(50, 56) SLN(9, 40, 41) DS(51, 56) SLN(10, 40, 41) DS(52, 56) SLN(11, 40, 41) DS(53, 56) SLN(12, 40, 41) DS(50, 56) SLN(13, 40, 41) DS(51, 56) SLN(14, 40, 41) DS(52, 56) SLN(15, 40, 41) DS(53, 56) "s_waitcnt lgkmcnt(0)\n\t" ::: CLOB, "v56");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: CLOB);
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    unsigned long long* cyc;
    (void)hipMalloc(&cyc, 8);
    const int iters = 2000;
    const char* names[] = {"VALU regs in other banks", "VALU reads MFMA's B register", "VALU sources in banks of A,B",
                           "VALU dst in bank of A", "VALU self-chain", "VALU writes MFMA's B register", "bare", "16 MFMA then 16 VALU", "4x(4 MFMA, 4 VALU)", "16x(MFMA, 2 VALU)", "1 VALU per 4 MFMA", "16x(MFMA, s_nop 0)", "16x(MFMA, ds_read_b32)"};
    for (int m = 0; m < 13; m++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (m) {
#define C(M) case M: k<M><<<256, 256>>>(cyc, iters); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
            }
            (void)hipDeviceSynchronize();
        }
        unsigned long long c;
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-34s %6.1f cyc/MFMA\n", names[m], c / (16.0 * iters));
    }
    return 0;
}
