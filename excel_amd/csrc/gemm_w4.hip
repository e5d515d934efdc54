// "w4" bf16x3 GEMM: the big tiles of gemm_bf16x3.hip on FOUR waves, one per SIMD, with a hand-placed instruction stream.
//
//   C[M,N] (fp32 or split) = act(A_split[M,K] . W_split[N,K]^T + bias) + residual          (same operands, layouts and epilogue semantics)
//
// Why a second kernel (rounds 3 and 4 measured this, EXPERIMENTS.md): the 8-wave kernel's k-loop takes 1.49x its own MFMA stream.  Its
// waves own 160 x 64 outputs - 28 fragment reads per 120 MFMAs - and all eight of them stop at a __syncthreads() every 32 k, read
// their B fragments and only then restart the matrix pipe.  Here a wave owns 16 NT_M x 128 outputs (NT_M = 10: 160 x 128):
//   * 36 ds_read_b128 per 240 MFMAs (0.15 instead of 0.23 LDS fragment bytes per MFMA: under the socket power cap every LDS byte
//     is clock), 320 accumulator registers = 256 AGPRs (row tiles 0-7) + 64 VGPRs (the two tail row tiles) of the 512-register budget
//     of a one-wave-per-SIMD kernel;
//   * the compiler only allocates registers: every instruction of the k-loop is an `asm volatile` statement (MFMA with an explicit
//     accumulator register class, ds_read_b128 with immediate offsets, counted s_waitcnt) or an LDS-DMA builtin between them, in
//     program order - hipcc would otherwise shuttle accumulator tiles between the two register files (340 v_accvgpr moves per
//     k-step, round 3) and drain the DMA queue in front of every LDS read it can see;
//   * ONE barrier per 32-k step, placed in front of the last two row tiles, and nothing waits behind it: by then every A fragment of
//     the step has been read (two tiles ahead, into a 4-deep register ring), so the barrier both publishes the next stage (each wave
//     waited for its own DMA pieces) and frees the current one.  The last two row tiles run column-pair-major, so the B fragments of a
//     column pair are dead after 12 MFMAs and are re-loaded from the NEXT stage right there; the next step's first two A tiles are
//     fetched at the head of this tail.  The matrix pipe never waits for an LDS round trip behind a barrier: by in-kernel stamps a step
//     of 240 MFMAs takes 4 300 cycles for 3 840 of matrix-pipe work (the bare one-wave MFMA stream: 4 080);
//   * MFMAs on one accumulator are 8 apart (pass-major over the 8 column tiles of a row tile; 4 apart in the tail): a single wave
//     has no partner to fill a dependent-accumulator wait (the first attempt, round 3: +44 % on the bare MFMA stream);
//   * the LDS-DMA pieces of the next stage (NT_M + 8 per wave) go out one at a time between MFMAs of the early row tiles, through a
//     buffer descriptor: loop-invariant 32-bit lane offsets + one scalar k offset (64-bit lane pointers spilled in round 3);
//   * the epilogue stores straight from the registers: the MFMA gets the weight fragment as srcA (the tile comes out transposed: a lane
//     holds columns of ONE row) and the weight rows are read in a permuted order with a swizzle of their own, so a lane owns 8
//     consecutive columns per tile pair (no LDS transpose, no barrier; see the epilogue).
// Instances: NT_M = 10 (320 x 256 tiles: the B = 32 layer shapes), 8 (256 x 256), 5 (160 x 256: N = 768 launches of the B = 16 shapes);
// excel_launch_gemm_bf16x3 picks instance vs 8-wave tile by modelled time.
// Stage layout: row = [hi 32 | lo 32] bf16 = 128 B = 8 chunks of 16 B; A rows (chunk c at slot c ^ ((row >> 1) & 7)), then the 256 B
// rows (slot c ^ swz_b(row)); two stages.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "excel_internal.h"


namespace EXCEL_SPLIT_NS {

#include "gemm_w4_body.inc"

// nt_m: 10 (320-row tiles), 8 (256) or 5 (160)
// x2: 0 = three MFMAs per product; 1 / 2 = the two-product instances of gemm_w4x2.hip (fp16-valued weights in the split layout / as a
// plain half matrix p.Bh)
bool excel_gemm_w4_supported(const GemmBfArgs& p, int nt_m, int x2) {
    const bool vec = (p.N & 3) == 0 && p.N >= 8 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 && (p.hd & 7) == 0;
    const int kq = 32 * ((x2 == 2 || (nt_m & 1)) ? 4 : 2);          // the k-loop is unrolled over 2 (4) steps of 32
    if (p.res && p.out_mode != GEMM_OUT_PLAIN) return false;      // the residual epilogue exists for the plain output only (all the path uses); else the 8-wave kernel
    if (x2 && !p.w_lo_zero) return false;
    if (x2 == 1 && nt_m == 10) return false;        // (not instantiated: gemm_w4x2.hip)
    if (x2 == 2 && (!p.Bh || p.ldbh < p.K || (p.ldbh & 7) || ((uintptr_t)p.Bh & 15) || (long long)p.N * p.ldbh * 2 >= 0x7fffffffLL)) return false;
    return (nt_m == 10 || nt_m == 8 || nt_m == 5) && vec && p.batch <= 1 && p.K >= kq && (p.K % kq) == 0 &&
           (long long)p.M * p.lda * 2 < 0x7fffffffLL && (long long)p.N * p.ldb * 2 < 0x7fffffffLL;
}

// Modelled time (us) of one launch of the nt_m instance on n_cu CUs: rounds of tiles (a partial last round counts less) x (prologue + row tiles x (k-steps x 0.233 + epilogue
// 1.8)); calibrated on the B = 32 layer shapes (profiles/r05_w4_arms.txt: a 320-row tile of K = 768 is 56 us of k-loop + 18 of epilogue + 7),
// the short instance pays ~8 % more per row tile for its fragment reads (26 instead of 36 per 240 MFMAs-equivalent).  The launcher compares
// this against the 8-wave tiles' model (gemm_bf16x3.hip).
// one tile of the nt_m instance (us): prologue + row tiles x (k-steps x 0.233 + epilogue 1.8)
static double w4_tile_us(int K, int nt_m, int x2) {
    // (two-product instances: 16 instead of 24 MFMAs per row tile and k-step)
    const double per_row_tile = (K / 32) * 0.233 * (x2 ? 0.70 : 1.0) * (nt_m == 5 ? 1.08 : nt_m == 8 ? 1.02 : 1.0) + 1.8;
    return 7.0 + nt_m * per_row_tile;
}
// `tiles` equal tiles on n_cu CUs, in tile-times: full rounds + a partly filled last round, which is cheaper than a full one (fewer CUs share
// the power budget and the fabric): 0.45 + 0.55 x fill, fitted on the B = 16 shapes (profiles/r05b_b16_shapes.txt: 360 tiles 140.6 us, 480
// tiles 163.1, 624 tiles 223.8)
static double w4_rounds(long long tiles, int n_cu) {
    if (tiles <= 0) return 0.0;
    const long long full = tiles / n_cu, rem = tiles - full * n_cu;
    return (double)full + (rem ? 0.45 + 0.55 * (double)rem / n_cu : 0.0);
}
double excel_gemm_w4_model_us(const GemmBfArgs& p, int nt_m, int n_cu, int x2) {
    return w4_rounds((long long)cdiv(p.M, 32 * nt_m) * cdiv(p.N, w4::BN), n_cu) * w4_tile_us(p.K, nt_m, x2);
}

// A launch made of TWO instances (gemm_w4_kernel_mix): R full rounds of 320-row tiles, the remaining rows in 256- or 160-row tiles that
// fill what is left of round R and (part of) one more.  -> modelled time, the split in *tall / *shrt (row tiles) and *second (8 / 5);
// 1e30 when no split applies.  (Judge, round 5: 711 tiles on 3 x 256 slots at B = 32, 1.4 - 2.4 rounds at B = 16.)
double excel_gemm_w4_mix_model_us(const GemmBfArgs& p, int n_cu, int x2, int* tall, int* shrt, int* second) {
    double best = 1e30;
    if (!excel_gemm_w4_supported(p, 10, x2)) return best;
    const int tiles_n = cdiv(p.N, w4::BN);
    const double t10 = w4_tile_us(p.K, 10, x2);
    const int cand[2] = {8, 5};
    for (int c = 0; c < 2; ++c) {
        if (!excel_gemm_w4_supported(p, cand[c], x2)) continue;
        const double ts = w4_tile_us(p.K, cand[c], x2);
        for (int R = 1; R <= 8; ++R) {
            const int a10 = (int)(((long long)R * n_cu) / tiles_n);
            if (a10 < 1 || (long long)a10 * 320 >= p.M) break;          // (the uniform grid covers M within R rounds)
            const int as = cdiv(p.M - a10 * 320, 32 * cand[c]);
            const long long slots_left = (long long)R * n_cu - (((long long)a10 * tiles_n + 7) & ~7LL);
            const double us = R * t10 + w4_rounds((long long)as * tiles_n - (slots_left > 0 ? slots_left : 0), n_cu) * ts;
            if (us < best) { best = us; *tall = a10; *shrt = as; *second = cand[c]; }
        }
    }
    return best;
}

// one plain __global__ function per instance (a kernel TEMPLATE launched from inside a function template lost its host-side stub)
#define W4_KERNEL(NT, DBG) __global__ __launch_bounds__(256, 1) void gemm_w4_kernel_##NT##_##DBG(GemmBfArgs p) { W4_UNIFORM_BODY(NT, DBG, 0); }
W4_KERNEL(10, 0) W4_KERNEL(8, 0) W4_KERNEL(5, 0)
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel_mix(GemmBfArgs p) { W4_MIX_BODY(0); }
#ifdef EXCEL_DEV
W4_KERNEL(10, 1) W4_KERNEL(10, 2) W4_KERNEL(10, 4) W4_KERNEL(10, 8) W4_KERNEL(8, 8) W4_KERNEL(5, 8) W4_KERNEL(10, 9) W4_KERNEL(10, 10)
W4_KERNEL(10, 15) W4_KERNEL(10, 24) W4_KERNEL(10, 32) W4_KERNEL(10, 136) W4_KERNEL(10, 143) W4_KERNEL(10, 128)
#endif
#define W4_LAUNCH(NT, DBG) hipLaunchKernelGGL(gemm_w4_kernel_##NT##_##DBG, grid, dim3(256), 0, stream, p)

static void launch_w4(const GemmBfArgs& p_in, int nt_m, hipStream_t stream) {
    GemmBfArgs p = p_in;
    const dim3 grid(cdiv(p.M, 32 * nt_m) * cdiv(p.N, w4::BN));
#ifdef EXCEL_DEV
    static const int stagger = getenv("EXCEL_W4_STAGGER") ? atoi(getenv("EXCEL_W4_STAGGER")) : 0;
    p.dbg = (stagger > 0 && (int)grid.x > 320) ? stagger : 0;      // multi-round launches only
    static const int dbg = getenv("EXCEL_W4_DBG") ? atoi(getenv("EXCEL_W4_DBG")) : 0;
    if (nt_m == 10) {
        switch (dbg) {
            case 1: W4_LAUNCH(10, 1); return;
            case 2: W4_LAUNCH(10, 2); return;
            case 4: W4_LAUNCH(10, 4); return;
            case 8: W4_LAUNCH(10, 8); return;
            case 9: W4_LAUNCH(10, 9); return;
            case 10: W4_LAUNCH(10, 10); return;
            case 15: W4_LAUNCH(10, 15); return;
            case 24: W4_LAUNCH(10, 24); return;
            case 32: W4_LAUNCH(10, 32); return;
            case 128: W4_LAUNCH(10, 128); return;
            case 136: W4_LAUNCH(10, 136); return;
            case 143: W4_LAUNCH(10, 143); return;
            default: break;
        }
    }
    if (dbg == 8 && nt_m == 8) { W4_LAUNCH(8, 8); return; }
    if (dbg == 8 && nt_m == 5) { W4_LAUNCH(5, 8); return; }
#endif
    if (nt_m == 10) W4_LAUNCH(10, 0);
    else if (nt_m == 8) W4_LAUNCH(8, 0);
    else W4_LAUNCH(5, 0);
}

int excel_launch_gemm_w4_mix(const GemmBfArgs& p_in, int tall, int shrt, int second, hipStream_t stream) {
    GemmBfArgs p = p_in;
    EXCEL_CHECK_ARG(excel_gemm_w4_supported(p, 10, 0) && excel_gemm_w4_supported(p, second, 0) && tall >= 1 && shrt >= 1 && (second == 8 || second == 5) &&
                    (long long)tall * 320 < p.M && (long long)tall * 320 + (long long)shrt * 32 * second >= p.M, "gemm_w4 (two instances): bad split");
    p.mix_tall = tall; p.mix_short = shrt; p.mix_first = second;
    const int tiles_n = cdiv(p.N, w4::BN);
    const dim3 grid(((tall * tiles_n + 7) & ~7) + shrt * tiles_n);
    hipLaunchKernelGGL(gemm_w4_kernel_mix, grid, dim3(256), 0, stream, p);
    EXCEL_CHECK_LAUNCH("gemm_w4 (two instances)");
    return EXCEL_OK;
}

int excel_launch_gemm_w4(const GemmBfArgs& p, int nt_m, hipStream_t stream) {
    EXCEL_CHECK_ARG(excel_gemm_w4_supported(p, nt_m, 0), "gemm_w4: unsupported problem (vector epilogue, batch 1, K %% 64 (128) == 0, operands below 2 GB)");
    launch_w4(p, nt_m, stream);
    EXCEL_CHECK_LAUNCH("gemm_w4");
    return EXCEL_OK;
}

}  // namespace EXCEL_SPLIT_NS

