// VALU issue-rate microbenchmark for MI355X (gfx950): how many wave64 VALU instructions per second the chip retires for the
// instruction kinds the traversal loop is made of.  hipcc --offload-arch=gfx950 -O3 valu_peak.hip -o valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    uint32_t u0 = threadIdx.x * 2654435761u;
    uint64_t m64 = 0x5555555555555555ull;
    for (int i = 0; i < iters; ++i) {
#define REP8(OP) OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
        if (KIND == 0) {
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            REP8(FMA) REP8(FMA)
        } else if (KIND == 1) {
#define CND(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a));
            REP8(CND) REP8(CND)
        } else if (KIND == 2) {
#define CVT(x) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x) : "v"(u0));
            REP8(CVT) REP8(CVT)
        } else if (KIND == 3) {
#define MX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            REP8(MX3) REP8(MX3)
        } else if (KIND == 4) {
#define PKF(x, y) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(y));
            double d0 = x0, d1 = x1, d2 = x2, d3 = x3, e = a;
            for (int r = 0; r < 4; ++r) { PKF(d0, e) PKF(d1, e) PKF(d2, e) PKF(d3, e) }
            x0 += (float)d0 + (float)d1 + (float)d2 + (float)d3;
        } else if (KIND == 5) {
#define RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
            REP8(RCP) REP8(RCP)
        } else if (KIND == 6) {
#define MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
            REP8(MUL) REP8(MUL)
        } else if (KIND == 7) {
#define CND64(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(m64));
            REP8(CND64) REP8(CND64)
        } else if (KIND == 8) {
#define MIN(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(a));
            REP8(MIN) REP8(MIN)
        } else if (KIND == 9) {
#define SUB(x) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(a));
            REP8(SUB) REP8(SUB)
        } else if (KIND == 10) {
#define CMP(x) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m64) : "v"(x), "v"(a));
            REP8(CMP) REP8(CMP)
        } else if (KIND == 11) {
#define AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(a));
            REP8(AND) REP8(AND)
        } else if (KIND == 12) {
#define MOV(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(a));
            REP8(MOV) REP8(MOV)
        } else if (KIND == 13) {
#define MINU(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(a));
            REP8(MINU) REP8(MINU)
        } else if (KIND == 14) {
#define PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            REP8(PERM) REP8(PERM)
        } else if (KIND == 15) {
#define LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(x) : "v"(a));
            REP8(LSHLOR) REP8(LSHLOR)
        } else if (KIND == 16) {
#define ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
            REP8(ADDU) REP8(ADDU)
        } else if (KIND == 17) {
#define MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            REP8(MED3) REP8(MED3)
        } else if (KIND == 19) {
#define FMX(x) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(u0), "v"(a));
            REP8(FMX) REP8(FMX)
        } else if (KIND == 20) {
#define BFE(x) asm volatile("v_bfe_u32 %0, %1, 5, 3" : "=v"(x) : "v"(u0));
            REP8(BFE) REP8(BFE)
        } else if (KIND == 21) {
#define MULLO(x) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x) : "v"(u0), "v"(u0));
            REP8(MULLO) REP8(MULLO)
        } else if (KIND == 18) {
#define CNDV(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
            REP8(CNDV)
        }
    }
    if (m64 == 12345) x0 += 1.f;
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int KIND> void run(const char *name, float *out, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 16, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waveInsts = (double)blocks * 4 * iters * 16;
    printf("%-18s %8.3f ms  %.3e wave-instructions/s  = %.2f per SIMD per 4 cycles at 2.4 GHz\n", name, ms, waveInsts / (ms * 1e-3), waveInsts / (ms * 1e-3) / (1024 * 2.4e9 / 4));
}
int main() {
    float *out; const int blocks = 256 * 8 * 4;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    run<0>("v_fma_f32", out, blocks); run<6>("v_mul_f32", out, blocks); run<1>("v_cndmask_b32", out, blocks); run<2>("v_cvt_f32_ubyte1", out, blocks);
    run<3>("v_max3_f32", out, blocks); run<5>("v_rcp_f32", out, blocks); run<4>("v_pk_fma_f32", out, blocks);
    run<7>("v_cndmask_e64 sgpr", out, blocks); run<8>("v_min_f32", out, blocks); run<9>("v_sub_f32", out, blocks); run<10>("v_cmp_lt_f32_e64", out, blocks);
    run<11>("v_and_b32", out, blocks); run<12>("v_mov_b32", out, blocks); run<13>("v_min_u32", out, blocks); run<14>("v_perm_b32", out, blocks);
    run<15>("v_lshl_or_b32", out, blocks); run<16>("v_add_u32", out, blocks); run<17>("v_med3_f32", out, blocks); run<18>("cmp+cndmask pair", out, blocks);
    run<19>("v_fma_mix_f32", out, blocks); run<20>("v_bfe_u32", out, blocks); run<21>("v_mul_lo_u32", out, blocks);
    return 0;
}
