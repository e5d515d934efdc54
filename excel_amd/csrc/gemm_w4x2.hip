// "f16x2": the four-wave GEMM of gemm_w4.hip for fp16-VALUED weights - two matrix-core instructions per product instead of three.
//
// Every published CLIP archive stores half-precision parameters; the reference loads them into an fp32 model unchanged
// (clip/build_model.py:72 keeps the fp32 conversion off, clip/clip.py:138-154), so in the IEEE-half split  w = hi + lo  the lo plane of
// every nn.Linear weight is exactly zero.  The three-MFMA product  a.w ~= al.wh + ah.wl + ah.wh  then carries one pass that multiplies
// zeros: it is dropped here.  The remaining two passes run in the order and on the accumulators of the f16x3 kernel, and adding a
// block of exact zeros to an fp32 accumulator does not change it - the results are BIT-IDENTICAL to f16x3 on the same operands
// (tests/test_gpu_ops.py::test_gemm_f16x2_equals_f16x3_bitwise), at two thirds of the matrix-pipe work of half of the step.
// Two instances per tile height (gemm_w4_body.inc):
//   X2 = 1: weights in the split layout (their zero lo halves ride along in the staged lines and are never read);
//   X2 = 2: weights as a plain half matrix [N][K] (excel_vit keeps one next to the split planes): half the weight bytes through L2, the
//           fabric and the LDS-DMA path - one 128-byte line of a weight row feeds two k-steps.
// Compiled for the IEEE-half split type only (build.py: F16_ONLY_SOURCES): a bf16 hi plane cannot hold an fp16 value.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "excel_internal.h"

#ifndef EXCEL_SPLIT_F16
#error "gemm_w4x2.hip is built for the IEEE-half split type only (-DEXCEL_SPLIT_F16)"
#endif

namespace EXCEL_SPLIT_NS {

#include "gemm_w4_body.inc"

#define W4X2_KERNEL(NT, X) __global__ __launch_bounds__(256, 1) void gemm_w4x2_kernel_##NT##_##X(GemmBfArgs p) { W4_UNIFORM_BODY(NT, 0, X); }
// (no 320-row instance on split-layout weights: its register allocation spills two accumulator tiles at the loop exit - the compact-weight
// instance below is the one the ViT runs; split-layout callers get the 256-row tile)
W4X2_KERNEL(8, 1) W4X2_KERNEL(5, 1)
W4X2_KERNEL(10, 2) W4X2_KERNEL(8, 2) W4X2_KERNEL(5, 2)
__global__ __launch_bounds__(256, 1) void gemm_w4x2_kernel_mix(GemmBfArgs p) { W4_MIX_BODY(2); }
#define W4X2_LAUNCH(NT, X) hipLaunchKernelGGL(gemm_w4x2_kernel_##NT##_##X, grid, dim3(256), 0, stream, p)
#ifdef EXCEL_DEV
// development arms of the 320-row compact-weight instance (EXCEL_W4_DBG, as in gemm_w4.hip): 1 no LDS-DMA after the prologue, 2 no fragment
// reads, 4 no barrier, 8 no epilogue, 16 no MFMAs, 128 cycle / phase stamps into the `bias` buffer (tools_dev/w4_stamps.py x2)
#define W4X2_KERNEL_D(DBG) __global__ __launch_bounds__(256, 1) void gemm_w4x2_kernel_10_2_d##DBG(GemmBfArgs p) { W4_UNIFORM_BODY(10, DBG, 2); }
W4X2_KERNEL_D(1) W4X2_KERNEL_D(2) W4X2_KERNEL_D(4) W4X2_KERNEL_D(8) W4X2_KERNEL_D(9) W4X2_KERNEL_D(10) W4X2_KERNEL_D(15) W4X2_KERNEL_D(128) W4X2_KERNEL_D(136) W4X2_KERNEL_D(143)
#define W4X2_LAUNCH_D(DBG) hipLaunchKernelGGL(gemm_w4x2_kernel_10_2_d##DBG, grid, dim3(256), 0, stream, p)
#endif

// x2 = 1 (split weights, p.B) or 2 (plain half weights, p.Bh / p.ldbh); preconditions: excel_gemm_w4_supported(p, nt_m, x2)
int excel_launch_gemm_w4x2(const GemmBfArgs& p, int nt_m, int x2, hipStream_t stream) {
    EXCEL_CHECK_ARG(p.w_lo_zero, "gemm_w4x2: the weight operand must be declared fp16-valued (w_lo_zero)");
    EXCEL_CHECK_ARG(excel_gemm_w4_supported(p, nt_m, x2), "gemm_w4x2: unsupported problem (vector epilogue, batch 1, K %% 64 (128) == 0, operands below 2 GB)");
    const dim3 grid(cdiv(p.M, 32 * nt_m) * cdiv(p.N, w4::BN));
#ifdef EXCEL_DEV
    static const int dbg = getenv("EXCEL_W4_DBG") ? atoi(getenv("EXCEL_W4_DBG")) : 0;
    if (x2 == 2 && nt_m == 10 && dbg) {
        switch (dbg) {
            case 1: W4X2_LAUNCH_D(1); break;
            case 2: W4X2_LAUNCH_D(2); break;
            case 4: W4X2_LAUNCH_D(4); break;
            case 8: W4X2_LAUNCH_D(8); break;
            case 9: W4X2_LAUNCH_D(9); break;
            case 10: W4X2_LAUNCH_D(10); break;
            case 15: W4X2_LAUNCH_D(15); break;
            case 128: W4X2_LAUNCH_D(128); break;
            case 136: W4X2_LAUNCH_D(136); break;
            default: W4X2_LAUNCH_D(143); break;
        }
        EXCEL_CHECK_LAUNCH("gemm_w4x2 (dev arm)");
        return EXCEL_OK;
    }
#endif
    if (x2 == 2) {
        if (nt_m == 10) W4X2_LAUNCH(10, 2);
        else if (nt_m == 8) W4X2_LAUNCH(8, 2);
        else W4X2_LAUNCH(5, 2);
    } else {
        if (nt_m == 8) W4X2_LAUNCH(8, 1);
        else W4X2_LAUNCH(5, 1);
    }
    EXCEL_CHECK_LAUNCH("gemm_w4x2");
    return EXCEL_OK;
}

// the two-instance launch of the compact-weight kernel (gemm_w4.hip: excel_gemm_w4_mix_model_us)
int excel_launch_gemm_w4x2_mix(const GemmBfArgs& p_in, int tall, int shrt, int second, hipStream_t stream) {
    GemmBfArgs p = p_in;
    EXCEL_CHECK_ARG(p.w_lo_zero && excel_gemm_w4_supported(p, 10, 2) && excel_gemm_w4_supported(p, second, 2) && tall >= 1 && shrt >= 1 &&
                    (second == 8 || second == 5) && (long long)tall * 320 < p.M && (long long)tall * 320 + (long long)shrt * 32 * second >= p.M,
                    "gemm_w4x2 (two instances): bad split");
    p.mix_tall = tall; p.mix_short = shrt; p.mix_first = second;
    const int tiles_n = cdiv(p.N, w4::BN);
    const dim3 grid(((tall * tiles_n + 7) & ~7) + shrt * tiles_n);
    hipLaunchKernelGGL(gemm_w4x2_kernel_mix, grid, dim3(256), 0, stream, p);
    EXCEL_CHECK_LAUNCH("gemm_w4x2 (two instances)");
    return EXCEL_OK;
}

}  // namespace EXCEL_SPLIT_NS
