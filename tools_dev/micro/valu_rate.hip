// dev micro-benchmark (gfx950): issue cost of the VALU / transcendental / packed / MFMA instruction mixes the attention strip kernel is
// made of, at 1 and 2 waves per SIMD.  One workgroup per CU-sized launch is enough: cycles are s_memtime deltas of wave 0.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = 0.999f, d = 1e-4f;
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    bf16x8 fa, fb;
    for (int q = 0; q < 8; ++q) { fa[q] = (__bf16)(0.001f * (threadIdx.x + q)); fb[q] = (__bf16)(0.002f * (threadIdx.x ^ q)); }
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {        // 64 independent-ish v_fma_f32 (8 chains)
            asm volatile(REP4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (MODE == 1) { // 64 v_pk_fma_f32 (4 chains)
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(f32x2{c, c}), "v"(f32x2{d, d}));
        } else if (MODE == 2) { // 64 v_exp_f32 (8 chains)
            asm volatile(REP4("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                              "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 3) { // 64 v_max3_f32
            asm volatile(REP4("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                              "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                              "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                              "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (MODE == 4) { // 16 x (exp + 3 fma): 64 instructions, the exps spaced out
            asm volatile(REP16("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (MODE == 5) { // 32 x (exp + fma)
            asm volatile(REP16("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (MODE == 6) { // 64 v_pk_add_f32
            asm volatile(REP16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(f32x2{d, d}));
        } else if (MODE == 7) { // 16 dependent MFMAs on one accumulator, nothing else
            asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n") : "+v"(acc0) : "v"(fa), "v"(fb));
        } else if (MODE == 8) { // 16 MFMAs alternating between two accumulators
            asm volatile(REP4("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n")
                         : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb));
        } else if (MODE == 9) { // 16 x (MFMA same accumulator + 4 fma fillers)
            asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_fma_f32 %3, %3, %7, %8\n v_fma_f32 %4, %4, %7, %8\n v_fma_f32 %5, %5, %7, %8\n v_fma_f32 %6, %6, %7, %8\n")
                         : "+v"(acc0) , "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
        } else if (MODE == 10) { // 16 x (MFMA alternating accumulators + 4 fma fillers)
            asm volatile(REP4("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              "v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              "v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(acc0), "+v"(acc1), "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
        } else if (MODE == 11) { // 16 x (MFMA same accumulator + 2 exp + 2 fma fillers)
            asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_exp_f32 %3, %3\n v_fma_f32 %4, %4, %7, %8\n v_exp_f32 %5, %5\n v_fma_f32 %6, %6, %7, %8\n")
                         : "+v"(acc0) , "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
        } else if (MODE == 12) { // 16 x (MFMA same accumulator + 8 fma fillers)
            asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_fma_f32 %3, %3, %7, %8\n v_fma_f32 %4, %4, %7, %8\n v_fma_f32 %5, %5, %7, %8\n v_fma_f32 %6, %6, %7, %8\n"
                               "v_fma_f32 %3, %3, %7, %8\n v_fma_f32 %4, %4, %7, %8\n v_fma_f32 %5, %5, %7, %8\n v_fma_f32 %6, %6, %7, %8\n")
                         : "+v"(acc0) , "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
        } else if (MODE == 13) { // the strip kernel's block shape: 12 chained MFMAs, then 48 VALU (16 exp, 16 fma, 16 add)
            asm volatile(REP4("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n")
                         REP16("v_fma_f32 %3, %3, %7, %8\n v_exp_f32 %4, %4\n v_add_f32 %5, %5, %8\n")
                         : "+v"(acc0) , "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
        } else if (MODE == 14) { // the same work with the VALU spread over the MFMA gaps (4 per gap)
            asm volatile(REP4("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_fma_f32 %3, %3, %7, %8\n v_exp_f32 %4, %4\n v_add_f32 %5, %5, %8\n v_fma_f32 %6, %6, %7, %8\n"
                              "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_exp_f32 %4, %4\n v_add_f32 %5, %5, %8\n v_fma_f32 %3, %3, %7, %8\n v_exp_f32 %6, %6\n"
                              "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_add_f32 %5, %5, %8\n v_fma_f32 %3, %3, %7, %8\n v_exp_f32 %4, %4\n v_add_f32 %6, %6, %8\n")
                         : "+v"(acc0) , "+v"(fa), "+v"(fb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    // the oldest wave of a SIMD wins every arbitration and never sees contention: report the slowest wave of the workgroup
    __shared__ long long tmax;
    if (threadIdx.x == 0) tmax = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)&tmax, (unsigned long long)(t1 - t0));
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = tmax;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + acc0[0] + acc1[3];
}

template <int MODE> void run(const char* name, int ninstr, float* out, long long* cyc) {
    for (int threads : {256, 512, 1024}) {     // 1, 2, 4 waves per SIMD (one workgroup per CU)
        const int iters = 200;
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        long long h[256];
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
        // s_memtime / readcyclecounter runs at a constant 100 MHz on this part: convert with the measured shader clock? report raw ticks per block too
        printf("%-58s waves/SIMD %d: %8.1f cycles per %d-instruction block (slowest wave), %6.2f per instruction per SIMD\n", name, threads / 256,
               s / 256 / iters, ninstr, s / 256 / iters / ninstr / (threads / 256));
    }
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 8);
    run<0>("64 v_fma_f32", 64, out, cyc);
    run<1>("64 v_pk_fma_f32", 64, out, cyc);
    run<6>("64 v_pk_add_f32", 64, out, cyc);
    run<2>("64 v_exp_f32", 64, out, cyc);
    run<3>("64 v_max3_f32", 64, out, cyc);
    run<4>("16 x (exp + 3 fma)", 64, out, cyc);
    run<5>("32 x (exp + fma)", 64, out, cyc);
    run<7>("16 MFMA one accumulator", 16, out, cyc);
    run<8>("16 MFMA two accumulators alternating", 16, out, cyc);
    run<9>("16 x (MFMA same acc + 4 fma)", 80, out, cyc);
    run<10>("16 x (MFMA alternating acc + 4 fma)", 80, out, cyc);
    run<11>("16 x (MFMA same acc + 2 exp + 2 fma)", 80, out, cyc);
    run<12>("16 x (MFMA same acc + 8 fma)", 144, out, cyc);
    run<13>("12 MFMA chained, then 16 x (fma, exp, add)", 60, out, cyc);
    run<14>("12 x (MFMA + 4 VALU of the same 48)", 60, out, cyc);
    return 0;
}
