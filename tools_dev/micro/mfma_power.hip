// dev micro-benchmark (gfx950): is the bf16 matrix pipe clock- / power-limited when the WHOLE chip runs the GEMM's MFMA stream, and does
// the operand data matter?  One 512-thread workgroup per CU (two waves per SIMD, as the bf16x3 GEMM), every wave runs `iters` blocks of
// 30 v_mfma_f32_32x32x16_bf16 on 10 accumulators (three dependent MFMAs per accumulator, the GEMM's order), operands in registers, no
// memory traffic.  Grid = 32 ... 256 workgroups; operand data = zeros / smooth values / random mantissas and exponents.
//   per SIMD: 2 waves x iters x 30 MFMAs x 32 cycles  ->  effective clock = cycles / time
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip && ./mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int F16, int ORDER>
__global__ __launch_bounds__(512) void k(float* out, int iters, int data) {
    f32x16 acc[10];
    for (int i = 0; i < 10; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    // eight operand fragments (the GEMM holds 2 B hi/lo + A hi/lo of two row tiles): distinct data per fragment
    unsigned short raw[8][8];
    for (int f = 0; f < 8; ++f)
        for (int q = 0; q < 8; ++q) {
            const unsigned h = hash32((blockIdx.x * 512 + threadIdx.x) * 64 + f * 8 + q);
            unsigned short v;
            if (data == 0) v = 0;
            else if (data == 1) v = F16 ? 0x2c00 + ((threadIdx.x + q) & 0xff) : 0x3c00 + ((threadIdx.x + q) & 0x3f);       // smooth: one exponent, low mantissa bits
            else v = F16 ? (unsigned short)(((h & 0x8000)) | (0x2400 + (h & 0x0fff))) : (unsigned short)((h & 0x8000) | (0x3a00 + (h & 0x03ff)));   // random sign / mantissa, 3-4 exponents
            raw[f][q] = v;
        }
    auto frag = [&](int f) {
        if constexpr (F16) { f16x8 x; __builtin_memcpy(&x, raw[f], 16); return x; }
        else { bf16x8 x; __builtin_memcpy(&x, raw[f], 16); return x; }
    };
    auto ah0 = frag(0), al0 = frag(1), ah1 = frag(2), al1 = frag(3), bh0 = frag(4), bl0 = frag(5), bh1 = frag(6), bl1 = frag(7);
    if constexpr (ORDER >= 4 && ORDER <= 8) {
        // the GEMM's real operand traffic: per 120 MFMAs (one 32-k stage of a wave) 28 ds_read_b128 refresh the fragments from an LDS tile of
        // random bits (ORDER 4), or only the 8 B fragments' worth (ORDER 5: what a four-wave kernel with twice the wave tile would read per
        // 120 MFMAs is 18).  No DMA, no barrier: LDS reads + MFMA only.
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        __shared__ __attribute__((aligned(16))) unsigned short tile[32 * 1024];          // 64 KB of random 16-bit patterns
        for (int i = threadIdx.x; i < 32 * 1024; i += 512) {
            const unsigned h = hash32(blockIdx.x * 65536 + i);
            tile[i] = data == 0 ? 0 : (unsigned short)((h & 0x8000) | (0x3a00 + (h & 0x03ff)));
        }
        __syncthreads();
        f32x4 a4[40];
        for (int i = 0; i < 40; ++i) a4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16x8* lt = reinterpret_cast<const bf16x8*>(tile);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        constexpr int NREAD = ORDER == 5 ? 18 : 28;
        if constexpr (ORDER >= 7) {
            // the shipped k-loop's shape: 8 B fragments at the head of a stage, A fragments streamed one 16-row tile ahead, a
            // sched_barrier after every row tile's 12 MFMAs (ORDER 8: plus the per-stage barrier)
            for (int it = 0; it < iters; ++it) {
                const int base = (it * 7 + wv * 8) * 64 + lane;
                bf16x8 bh[4], bl[4], ah[2], al[2];
#pragma unroll
                for (int j = 0; j < 4; ++j) { bh[j] = lt[(base + j * 128) & 4095]; bl[j] = lt[(base + j * 128 + 64) & 4095]; }
                ah[0] = lt[(base + 1024) & 4095]; al[0] = lt[(base + 1088) & 4095];
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    if (i + 1 < 10) { ah[(i + 1) & 1] = lt[(base + 1024 + (i + 1) * 128) & 4095]; al[(i + 1) & 1] = lt[(base + 1088 + (i + 1) * 128) & 4095]; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i & 1], bh[j], a4[i * 4 + j], 0, 0, 0);
                        a4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bl[j], a4[i * 4 + j], 0, 0, 0);
                        a4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bh[j], a4[i * 4 + j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (ORDER == 8) __syncthreads();
            }
        } else
        for (int it = 0; it < iters; ++it) {
            bf16x8 fr[28];
#pragma unroll
            for (int q = 0; q < NREAD; ++q) fr[q] = lt[((it * 7 + wv * 8) * 64 + lane + q * 128) & 4095];      // 28 distinct 1-KB fragments
#pragma unroll
            for (int q = NREAD; q < 28; ++q) fr[q] = fr[q - NREAD];
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[8 + 2 * i], fr[j], a4[i * 4 + j], 0, 0, 0);
                    a4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[9 + 2 * i], fr[4 + j], a4[i * 4 + j], 0, 0, 0);
                    a4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[9 + 2 * i], fr[j], a4[i * 4 + j], 0, 0, 0);
                }
            if constexpr (ORDER == 6) __syncthreads();
        }
        float s4 = 0.f;
        for (int i = 0; i < 40; ++i) s4 += a4[i][0] + a4[i][1] + a4[i][2] + a4[i][3];
        if (s4 == 1.2345e-30f) out[0] = s4;
        return;
    }
    if constexpr (ORDER == 3) {
        // 16x16x32 with the accumulators in AGPRs (inline asm, "+a"): does the accumulator file matter for power?
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 a4[20];
        for (int i = 0; i < 20; ++i) a4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                const auto& ah = (i & 1) ? ah1 : ah0;
                const auto& al = (i & 1) ? al1 : al0;
                const auto& bh = (i & 2) ? bh1 : bh0;
                const auto& bl = (i & 2) ? bl1 : bl0;
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n v_mfma_f32_16x16x32_bf16 %0, %3, %2, %0"
                             : "+a"(a4[i]) : "v"(al), "v"(bh), "v"(ah), "v"(bl));
            }
        }
        float s4 = 0.f;
        for (int i = 0; i < 20; ++i) s4 += a4[i][0] + a4[i][1] + a4[i][2] + a4[i][3];
        if (s4 == 1.2345e-30f) out[0] = s4;
        return;
    }
    if constexpr (ORDER == 2) {
        // the same flops as 16x16x32 instructions (60 per block on 20 accumulators of 4 registers, three dependent per accumulator)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 a4[20];
        for (int i = 0; i < 20; ++i) a4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                const auto& ah = (i & 1) ? ah1 : ah0;
                const auto& al = (i & 1) ? al1 : al0;
                const auto& bh = (i & 2) ? bh1 : bh0;
                const auto& bl = (i & 2) ? bl1 : bl0;
                if constexpr (!F16) {
                    a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, a4[i], 0, 0, 0);
                    a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, a4[i], 0, 0, 0);
                    a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, a4[i], 0, 0, 0);
                } else {
                    a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, a4[i], 0, 0, 0);
                    a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, a4[i], 0, 0, 0);
                    a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, a4[i], 0, 0, 0);
                }
            }
        }
        float s4 = 0.f;
        for (int i = 0; i < 20; ++i) s4 += a4[i][0] + a4[i][1] + a4[i][2] + a4[i][3];
        if (s4 == 1.2345e-30f) out[0] = s4;
        return;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const auto& ah = (i & 1) ? ah1 : ah0;
            const auto& al = (i & 1) ? al1 : al0;
            if constexpr (ORDER == 1 && !F16) {
                // operand-reuse order: A changes twice, B five times per six MFMAs (shipped order: 4 + 6); the three products of an accumulator
                // are still al.bh, ah.bh, ah.bl but interleaved over the two accumulators of the row tile
                acc[i * 2 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0, acc[i * 2 + 0], 0, 0, 0);
                acc[i * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1, acc[i * 2 + 1], 0, 0, 0);
                acc[i * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc[i * 2 + 1], 0, 0, 0);
                acc[i * 2 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc[i * 2 + 0], 0, 0, 0);
                acc[i * 2 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0, acc[i * 2 + 0], 0, 0, 0);
                acc[i * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1, acc[i * 2 + 1], 0, 0, 0);
                continue;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const auto& bh = j ? bh1 : bh0;
                const auto& bl = j ? bl1 : bl0;
                if constexpr (F16) {
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i * 2 + j], 0, 0, 0);
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i * 2 + j], 0, 0, 0);
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i * 2 + j], 0, 0, 0);
                } else {
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i * 2 + j], 0, 0, 0);
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i * 2 + j], 0, 0, 0);
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i * 2 + j], 0, 0, 0);
                }
            }
        }
        // keep the accumulators bounded (random data would overflow over thousands of blocks): nothing - fp32 inf is still an MFMA
    }
    float s = 0.f;
    for (int i = 0; i < 10; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 1.2345e-30f) out[0] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;       // 2000 x 30 x 32 x 2 = 3.84 M cycles per SIMD (~2 ms)
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* dn[3] = {"zeros ", "smooth", "random"};
    const int norder = argc > 2 ? 9 : 1;                      // any second argument: also the operand-reuse order and the 16x16x32 form (bf16, random data)
    for (int order = 0; order < norder; ++order)
    for (int f16 = 0; f16 < ((order == 1 || order >= 3) ? 1 : 2); ++f16)
        for (int data = (order ? 2 : 0); data < 3; ++data)
            for (int grid : {32, 64, 128, 192, 256}) {
                float best = 1e30f, last = 0.f;
                for (int rep = 0; rep < 6; ++rep) {              // back to back: the later repetitions see the steady-state clock
                    hipEventRecord(e0);
                    const int its = order >= 4 ? iters / 2 : iters;     // (a block of the LDS arms is 120 MFMAs of 16 cycles = two 30 x 32-cycle blocks)
                    if (order == 8) hipLaunchKernelGGL((k<0, 8>), dim3(grid), dim3(512), 0, 0, out, its, data);
                    else if (order == 7) hipLaunchKernelGGL((k<0, 7>), dim3(grid), dim3(512), 0, 0, out, its, data);
                    else if (order == 6) hipLaunchKernelGGL((k<0, 6>), dim3(grid), dim3(512), 0, 0, out, its, data);
                    else if (order == 5) hipLaunchKernelGGL((k<0, 5>), dim3(grid), dim3(512), 0, 0, out, its, data);
                    else if (order == 4) hipLaunchKernelGGL((k<0, 4>), dim3(grid), dim3(512), 0, 0, out, its, data);
                    else if (order == 3) hipLaunchKernelGGL((k<0, 3>), dim3(grid), dim3(512), 0, 0, out, iters, data);
                    else if (order == 2 && f16) hipLaunchKernelGGL((k<1, 2>), dim3(grid), dim3(512), 0, 0, out, iters, data);
                    else if (order == 2) hipLaunchKernelGGL((k<0, 2>), dim3(grid), dim3(512), 0, 0, out, iters, data);
                    else if (order) hipLaunchKernelGGL((k<0, 1>), dim3(grid), dim3(512), 0, 0, out, iters, data);
                    else if (f16) hipLaunchKernelGGL((k<1, 0>), dim3(grid), dim3(512), 0, 0, out, iters, data);
                    else hipLaunchKernelGGL((k<0, 0>), dim3(grid), dim3(512), 0, 0, out, iters, data);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&last, e0, e1);
                    if (last < best) best = last;
                }
                const double cyc = 2.0 * iters * 30 * 32;
                printf("%s%s %s grid %3d: best %8.1f us  last %8.1f us  -> %.2f GHz effective (last), %.0f TFLOP/s dense-equivalent\n", order == 8 ? "16x16x32 kernel-shaped reads + barrier " : order == 7 ? "16x16x32 kernel-shaped reads " : order == 6 ? "16x16x32+28 LDS reads/120 + barrier " : order == 5 ? "16x16x32+18 LDS reads/120 " : order == 4 ? "16x16x32+28 LDS reads/120 " : order == 3 ? "16x16x32-agpr " : order == 2 ? "16x16x32 " : order ? "reuse-order " : "", f16 ? "f16 " : "bf16", dn[data], grid,
                       best * 1e3, last * 1e3, cyc / (last * 1e-3) * 1e-9, grid * 8.0 * iters * 30 * 32768.0 / (last * 1e-3) * 1e-12);
            }
    return 0;
}
