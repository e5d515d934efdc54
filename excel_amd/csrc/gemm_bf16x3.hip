// "bf16x3" GEMM: fp32-grade GEMM on the CDNA4 bf16 matrix core.
//
// Every fp32 operand x is carried as two bf16 planes  x = hi + lo  (hi = bf16(x), lo = bf16(x - hi); 16 mantissa bits,
// fp32 exponent range, so no scaling and no overflow concerns) and a product is evaluated as
//     a.b ~= ah.bh + ah.bl + al.bh          (the al.bl term is 2^-16 relative and dropped)
// with three matrix-core instructions into ONE fp32 accumulator - since round 4 v_mfma_f32_16x16x32_bf16 (K = 32 per instruction: half the
// accumulator updates per flop of the 32x32x16 form; under the chip's power cap it sustains 2 130 instead of 1 800 TFLOP/s on random bits).  Products of bf16 values are exact in fp32, so the
// only error is the dropped term and the lo truncation: ~2^-17 relative per product.  Measured through the whole
// ViT-B/16 surgery forward at 448^2 the CAM moves by 8.8e-6 max-abs against exact fp32 (gate: 1e-3) -- see DESIGN.md 2.
// The bf16 pipe issues 16x the MACs per cycle of the f32-input MFMA, so 3 MFMAs per product is a 5.3x higher ceiling
// (2.5 PFLOP/s / 3 = 833 TFLOP/s fp32-equivalent vs 157 TFLOP/s).
//
// "Split" tensors hold 2K bf16 per row, per block of 32 k: 32 hi values then 32 lo values (split_off, common.h), so one
// GEMM k-step reads ONE full 128-B line per row (fetching hi and lo planes separately halves the useful bytes per L2
// line and measured ~2x the L2 traffic): exactly the bytes of the fp32
// tensor, so they live in the same workspace buffers.  Producers write them directly (LayerNorm, the QuickGELU
// epilogue of this kernel, the attention output, im2col); weights are split once at excel_vit_create.
//
//   C[M,N] (fp32 or split) = act(A_split[M,K] . W_split[N,K]^T + bias) + residual
// Tiling: 128x128x32 block tile, 4 waves (2x2) x (2x2) MFMA tiles of 32x32, LDS double buffer,
// direct-to-LDS staging
// (global_load_lds_dwordx4, XOR-swizzled 128-B rows: conflict-free ds_read_b128), XCD-aware tile order; per 32-k stage a wave of the
// 128 x 128 tile does 16 ds_read_b128 and 48 MFMAs (16 x 16 x 32).
#include <stdlib.h>
#include "common.h"
#include "excel_internal.h"

namespace EXCEL_SPLIT_NS {     // compiled once per 16-bit split type (excel_internal.h, build.py)

typedef unsigned short u16;

#define TBM 128
#define TBN 128
#define TBK 32
#define TROW 64     // bf16 elements per LDS row: [hi 32 | lo 32] = 128 B = 8 chunks of 16 B, XOR-swizzled (no padding)

// Shared epilogue of the GEMM kernels: bias -> activation -> (+ residual) -> store, specialised per output mode so the
// 64 accumulator elements of a thread see no per-element mode branches, integer divisions or 64-bit multiplies.
template <int OUT_MODE, int TI>
__device__ __forceinline__ void bf_epilogue(const GemmBfArgs& p, const f32x4 (&acc)[2 * TI][4], int m0, int n0, int wm, int wn, int lane, int nti = TI) {
    const int r = lane & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + r;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
        long long qcol_off = 0;
        int qd = 0;
        if (OUT_MODE == GEMM_OUT_QKV_HEADMAJOR) {
            const int D = p.heads * p.hd;
            const int qt = col / D, rem = col - qt * D;
            const int qh = rem / p.hd;
            qd = rem - qh * p.hd;
            qcol_off = ((long long)qt * p.heads + qh) * p.tokN * p.hd + qd;     // + (b*3*heads*tokN + n) * hd per row
        }
#pragma unroll
        for (int i = 0; i < 2 * TI; ++i) {
            if (i >= 2 * nti) continue;                           // (mixed-height tiles: a short tile's waves own nti = TI - 1 row tiles, and their rows start at wm * nti * 32)
            const int rbase = m0 + wm * (nti * 32) + i * 16 + 4 * (lane >> 4);
            int qb = 0, qn = 0;
            if (OUT_MODE == GEMM_OUT_QKV_HEADMAJOR) { qb = rbase / p.tokN; qn = rbase - qb * p.tokN; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int dr = e;
                const int row = rbase + dr;
                if (row >= p.M) continue;
                float v = acc[i][j][e] + bv;
                if (p.act == GEMM_ACT_QUICKGELU) v = v * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v));
                if (p.res) v += p.res[(long long)row * p.ldr + col];
                if (OUT_MODE == GEMM_OUT_SPLIT_BF16) {
                    const split_t hi = split_hi(v);
                    split_t* o = reinterpret_cast<split_t*>(p.Cs) + (long long)row * 2 * p.N + split_off(col, 0);
                    o[0] = hi;
                    o[32] = split_hi(v - (float)hi);
                } else if (OUT_MODE == GEMM_OUT_QKV_HEADMAJOR) {
                    int n = qn + dr, b = qb;
                    while (n >= p.tokN) { n -= p.tokN; ++b; }
                    const long long off = qcol_off + ((long long)b * 3 * p.heads * p.tokN + n) * p.hd;
                    if (!p.qkv_split) p.C[off] = v;
                    if (p.qkv_split) {     // same (b,type,head,n) row, [hi hd | lo hd] bf16: operand of the bf16x3 attention scores
                        split_t* o = reinterpret_cast<split_t*>(p.qkv_split) + (off - qd) * 2 + qd;
                        const split_t hi = split_hi(v);
                        o[0] = hi;
                        o[p.hd] = split_hi(v - (float)hi);
                    }
                } else {
                    p.C[(long long)row * p.ldc + col] = v;
                }
            }
        }
    }
}

// LDS-staged epilogue: the accumulator layout (one column per lane, rows spread over registers) makes direct stores
// 4-byte scatters (two 128-B segments per instruction; measured 28 % of the fc1 GEMM).  Each wave instead transposes its
// 64x64 tile through 2 x (32 x 68-float) LDS rounds and stores row-contiguous 16-byte vectors: bias / residual become
// float4 loads, fp32 rows are written as full 256-B segments, split-bf16 rows as 8-byte hi/lo groups.
template <int OUT_MODE, int TI>
__device__ __forceinline__ void bf_epilogue_lds(const GemmBfArgs& p, const f32x4 (&acc)[2 * TI][4], int m0, int n0, int wm, int wn,
                                                int lane, float* scratch, int nti = TI) {
    const int r = lane & 15;
    const int c4 = (lane & 15) * 4;
    const int col = n0 + wn * 64 + c4;
    const bool col_ok = col < p.N;                       // N % 4 == 0: the whole float4 is inside or outside
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && col_ok) bias4 = *reinterpret_cast<const f32x4*>(p.bias + col);
    int qt = 0, qh = 0, qd = 0;
    if (OUT_MODE == GEMM_OUT_QKV_HEADMAJOR) {
        const int D = p.heads * p.hd;
        qt = col / D;
        const int rem = col - qt * D;
        qh = rem / p.hd;
        qd = rem - qh * p.hd;
    }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        if (i >= nti) break;                                  // (mixed-height tiles: a short tile has nti = TI - 1 row tiles per wave)
        // accumulator tile (16-row tile t, 16-column tile j): register e = row 4 (lane / 16) + e, column lane % 16
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { if (EXCEL_DBG(p.dbg) & 32) break; scratch[(t * 16 + 4 * (lane >> 4) + e) * 68 + j * 16 + r] = acc[2 * i + t][j][e]; }
        __builtin_amdgcn_s_waitcnt(0xc07f);               // this wave's LDS writes landed (same-wave read back)
        // Batched by hand: 8 LDS reads, then (residual) 8 global loads, then the math, then the stores.  Written as one loop with the row
        // test in front, every iteration was its own chain of basic blocks - ds_read, lgkmcnt(0), residual load, vmcnt(0) (which also waits
        // for the previous iteration's stores), stores - 40 times in series per wave.
        constexpr int EB = 4;                              // iterations per batch (8 at once spill: the later row tiles' accumulators are still live)
#pragma unroll
        for (int h0 = 0; h0 < 8; h0 += EB) {
        f32x4 v[EB];
#pragma unroll
        for (int it = 0; it < EB; ++it) {
            const int row_l = (h0 + it) * 4 + (lane >> 4);
            if (EXCEL_DBG(p.dbg) & 32) { v[it][0] = acc[2 * i][it][0]; v[it][1] = acc[2 * i][it][1]; v[it][2] = acc[2 * i + 1][it][2]; v[it][3] = acc[2 * i + 1][it][3]; }   // dev arm: no LDS transpose (wrong values, same registers live)
            else v[it] = *reinterpret_cast<const f32x4*>(&scratch[row_l * 68 + c4]);
        }
        const int row0 = m0 + wm * (nti * 32) + i * 32 + (lane >> 4);
        const int colc = col_ok ? col : 0;
        int qb0 = 0, qn0 = 0;
        if (OUT_MODE == GEMM_OUT_QKV_HEADMAJOR) { qb0 = row0 / p.tokN; qn0 = row0 - qb0 * p.tokN; }
        if (p.res) {
            f32x4 rs[EB];
#pragma unroll
            for (int it = 0; it < EB; ++it) rs[it] = *reinterpret_cast<const f32x4*>(p.res + (long long)min(row0 + (h0 + it) * 4, p.M - 1) * p.ldr + colc);
#pragma unroll
            for (int it = 0; it < EB; ++it) v[it] += bias4;
            if (p.act == GEMM_ACT_QUICKGELU) {
#pragma unroll
                for (int it = 0; it < EB; ++it)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[it][q] = v[it][q] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[it][q]));
            }
#pragma unroll
            for (int it = 0; it < EB; ++it) v[it] += rs[it];
        } else {
#pragma unroll
            for (int it = 0; it < EB; ++it) v[it] += bias4;
            if (p.act == GEMM_ACT_QUICKGELU) {
#pragma unroll
                for (int it = 0; it < EB; ++it)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[it][q] = v[it][q] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[it][q]));
            }
        }
#pragma unroll
        for (int it = 0; it < EB; ++it) {
            const int row = row0 + (h0 + it) * 4;
            if ((EXCEL_DBG(p.dbg) & 16) && v[it][0] + v[it][1] + v[it][2] + v[it][3] != 1.2345e-30f) continue;          // dev arm: no global stores
            if (row >= p.M || !col_ok) continue;
            if (OUT_MODE == GEMM_OUT_PLAIN) {
                *reinterpret_cast<f32x4*>(p.C + (long long)row * p.ldc + col) = v[it];
            } else {
                split_t hi[4], lo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { hi[q] = split_hi(v[it][q]); lo[q] = split_hi(v[it][q] - (float)hi[q]); }
                if (OUT_MODE == GEMM_OUT_SPLIT_BF16) {
                    split_t* o = reinterpret_cast<split_t*>(p.Cs) + (long long)row * 2 * p.N + split_off(col, 0);
                    *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(hi);
                    *reinterpret_cast<uint2*>(o + 32) = *reinterpret_cast<const uint2*>(lo);
                } else {   // q|k|v head-major: fp32 for P.V / A.V, and [hi hd | lo hd] bf16 for the bf16x3 scores
                    int b = qb0, n = qn0 + (h0 + it) * 4;                     // (one division per row tile, not per store)
                    while (n >= p.tokN) { n -= p.tokN; ++b; }
                    const long long rowidx = (((long long)b * 3 + qt) * p.heads + qh) * p.tokN + n;
                    // the fp32 copy is not consumed when the split copy exists (bf16x3 attention; V^T is cut from the split planes)
                    if (!p.qkv_split) *reinterpret_cast<f32x4*>(p.C + rowidx * p.hd + qd) = v[it];
                    if (p.qkv_split) {
                        split_t* o = reinterpret_cast<split_t*>(p.qkv_split) + rowidx * 2 * p.hd + qd;
                        *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(hi);
                        *reinterpret_cast<uint2*>(o + p.hd) = *reinterpret_cast<const uint2*>(lo);
                    }
                }
            }
        }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);               // reads done before the next round overwrites the scratch
    }
}

// WM x WN waves, each owning TI x 2 MFMA tiles (TI*32 rows x 64 columns): block tile = (WM*TI*32) x (WN*64);
// NSTAGE-deep LDS ring filled by global_load_lds.  Inside this kernel the staging path (L2 -> LDS) delivered ~27 B/clk per CU whatever
// the tile (round 1; the port itself sustains 65 B/clk per CU with nothing else running, tools_dev/micro/ldsdma_rate.hip), so the lever is
// bytes per flop:
//   <2,2,2,2>: 128x128, 64 KB LDS, 2 workgroups/CU                      (small / batched problems)
//   <4,2,2,3>: 256x128, 144 KB, 8 waves, counted vmcnt ring             (N = 768 GEMMs: keeps 594 tiles for 256 CUs)
//   <2,4,4,2>: 256x256, 128 KB, 8 waves x (128x64), half the bytes/flop of 128x128   (QKV, fc1)
// MIX (320 x 256 instance only): MIXED-HEIGHT row tiles.  M = 25 120 rows are 785 blocks of 32; uniform 320-row tiles give every CU
// 3 (QKV: 9 column tiles) or 4 (fc1: 12) tiles of 10 blocks = 30 / 40 where 27.6 / 36.8 would do (92 % of the rounds filled).  With
// `mix_tall` row tiles of 320 rows followed by `mix_short` row tiles of 256 rows (the same workgroup shape; a short tile's waves run 4
// of their 5 row tiles) the same number of tiles covers M with less padding: a CU gets 2 tall + 1 short = 28 (3 + 1 = 38).  Tall tiles
// are dispatched first inside every XCD's chunk (longest first), the short ones level the last round.
// X2 ("f16x2", IEEE-half split type only): the weight operand is fp16-valued - its lo plane is all zero (GemmBfArgs::w_lo_zero) - so the
// a.hi x b.lo instruction of every fragment pair multiplies zeros and is left out, together with the b.lo fragment reads: two MFMAs per
// product, the bits of the three-product kernel (x + 0 = x in the fp32 accumulator).
template <int WM, int WN, int TI, int NSTAGE, bool MIX = false, bool X2 = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_bf16x3_kernel(GemmBfArgs p) {
    constexpr int BM = WM * TI * 32, BN = WN * 64;
    constexpr int STAGE = (BM + BN) * TROW;                  // u16 elements per stage: A rows then B rows
    constexpr int NWAVE = WM * WN;
    constexpr int A_PER_WAVE = (BM / 8) / NWAVE;             // 1-KB row groups (8 rows) each wave stages per k-step
    constexpr int B_PER_WAVE = (BN / 8) / NWAVE;
    constexpr int PER_WAVE = A_PER_WAVE + B_PER_WAVE;
    static_assert(A_PER_WAVE * NWAVE * 8 == BM && B_PER_WAVE * NWAVE * 8 == BN, "tile rows must split evenly over the waves");
    // each row is 8 x 16-B chunks [hi 32 | lo 32], chunk c stored at slot c ^ ((row>>1)&7)
    __shared__ __attribute__((aligned(1024))) u16 smem[NSTAGE * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, kh = lane >> 5;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    int m0, n0, nti = TI;
    if (MIX) {
        // XCD x (hardware id & 7) owns a contiguous run of the tall tiles and a contiguous run of the short tiles, tall ones first
        const int T = p.mix_tall * tiles_n, A = (p.mix_tall + p.mix_short) * tiles_n;
        const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
        const int tq = T >> 3, tr = T & 7, aq = A >> 3, ar = A & 7;
        const int tbase = x * tq + min(x, tr), tcount = tq + (x < tr ? 1 : 0);
        const int abase = x * aq + min(x, ar);
        const int first = min(p.mix_first, aq + (x < ar ? 1 : 0) - tcount);     // short tiles of this XCD that go out before its tall ones
        int rt, tn;
        if (loc >= first && loc < first + tcount) {
            const int k = tbase + (loc - first);
            rt = k / tiles_n; tn = k - rt * tiles_n;
            m0 = rt * BM;
        } else {
            const int k = (abase - tbase) + (loc < first ? loc : loc - tcount);   // index among the short tiles
            rt = k / tiles_n; tn = k - rt * tiles_n;
            m0 = p.mix_tall * BM + rt * (BM - 64);
            nti = TI - 1;
        }
        n0 = tn * BN;
    } else {
        const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
        const int tm = id / tiles_n, tn = id % tiles_n;
        m0 = tm * BM; n0 = tn * BN;
    }
    const int wm_rows = nti * 32;                            // rows per wave row of this tile
    if (p.batch > 1) {      // batched problems (blockIdx.y): advance the operand / output bases (block-uniform)
        const long long z = blockIdx.y;
        p.A += z * p.sA; p.B += z * p.sB; p.C += z * p.sC; p.Cs += z * p.sCs;
    }

    // direct-to-LDS staging (global_load_lds_dwordx4): one wave-instruction fills 1 KB = 8 tile rows; the LDS image is
    // lane-linear, so the swizzle is applied to the per-lane SOURCE address.
    const u16* gsrc[PER_WAVE];
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const bool isB = j >= A_PER_WAVE;
        const int seg = isB ? wave * B_PER_WAVE + (j - A_PER_WAVE) : wave * A_PER_WAVE + j;
        const int row_l = seg * 8 + (lane >> 3), phys = lane & 7;
        const int c = phys ^ ((row_l >> 1) & 7);
        const int grow = isB ? min(n0 + row_l, p.N - 1) : min(m0 + row_l, p.M - 1);
        gsrc[j] = (isB ? p.B + (long long)grow * p.ldb : p.A + (long long)grow * p.lda) + c * 8;   // chunk c of the row's 128-B k-block
    }
    auto issue_piece = [&](int kt, int stage, int j) {
        const bool isB = j >= A_PER_WAVE;
        const int seg = isB ? wave * B_PER_WAVE + (j - A_PER_WAVE) : wave * A_PER_WAVE + j;
        u16* dst = smem + stage * STAGE + (isB ? BM * TROW : 0) + seg * 8 * TROW;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j] + kt * 64),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto issue_tile = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) issue_piece(kt, stage, j);
    };

    // accumulators: 2 TI x 4 tiles of 16 x 16 (v_mfma_f32_16x16x32: one instruction covers the whole 32-k stage)
    f32x4 acc[2 * TI][4];
#pragma unroll
    for (int i = 0; i < 2 * TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, kg = lane >> 4;     // fragment row / column of a 16-wide tile, k group (8 k each)
    const int sw = (r16 >> 1) & 7;    // (row>>1)&7 of every fragment row this lane reads (tile rows are r16 + multiples of 16)
    const int nk = p.K / TBK;

    // kt_next >= 0 (TI == 5 path): the next tile's LDS-DMA pieces are issued one per (k16 sub-step, row tile) pair INSIDE the MFMA
    // stream instead of all PER_WAVE up front.  A piece occupies the CU's one address / L1 path for ~16 cycles (64 lanes x 16 B at
    // 64 B/clk): with all 8 waves issuing 9 pieces at the step start that queue was ~1 100 cycles deep, and a wave does not reach its
    // MFMAs before its own pieces are accepted - measured (PMC, fc2 shape): +75 k of 572 k kernel cycles wherever the data came from.
    // Same-box A/B of placements (tools_dev/gemm_bench.py, four layer shapes): one piece at the head of each pair -4.5 %; two per
    // second pair -1 %; behind the pair's MFMAs -3 %; the two waves of a SIMD offset by one pair -3 %; a branch-free peeled loop -2.4 %.
    auto compute = [&](int stage, int kt_next) {
        const u16* as = smem + stage * STAGE;
        const u16* bs = as + BM * TROW;
        const int ch = (kg ^ sw) * 8, cl = ((4 + kg) ^ sw) * 8;       // this lane's 16-byte chunk of the hi / lo half of a 128-byte row
        splitx8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u16* rowp = bs + (wn * 64 + j * 16 + r16) * TROW;
            bh[j] = *reinterpret_cast<const splitx8*>(rowp + ch);
            if (!X2) bl[j] = *reinterpret_cast<const splitx8*>(rowp + cl);
        }
        // A fragments are streamed one 16-row tile ahead (the accumulators leave < 100 registers for everything else in the 320 x 256
        // tile), and the scheduler may not hoist them
        splitx8 ah[2], al[2];
        {
            const u16* rowp = as + (wm * wm_rows + r16) * TROW;
            ah[0] = *reinterpret_cast<const splitx8*>(rowp + ch);
            al[0] = *reinterpret_cast<const splitx8*>(rowp + cl);
        }
#pragma unroll
        for (int i = 0; i < 2 * TI; ++i) {
            if (kt_next >= 0 && i < PER_WAVE) issue_piece(kt_next, stage ^ 1, i);
            if (MIX && i >= 2 * nti) continue;                  // short tile: this wave has 2 (TI - 1) row tiles (the DMA piece above still goes out)
            if (i + 1 < 2 * TI && (!MIX || i + 1 < 2 * nti)) {
                const u16* rowp = as + (wm * wm_rows + (i + 1) * 16 + r16) * TROW;
                ah[(i + 1) & 1] = *reinterpret_cast<const splitx8*>(rowp + ch);
                al[(i + 1) & 1] = *reinterpret_cast<const splitx8*>(rowp + cl);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // small terms first, the dominant hi.hi product last
                acc[i][j] = EXCEL_MFMA32(al[i & 1], bh[j], acc[i][j]);
                if (!X2) acc[i][j] = EXCEL_MFMA32(ah[i & 1], bl[j], acc[i][j]);
                acc[i][j] = EXCEL_MFMA32(ah[i & 1], bh[j], acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (EXCEL_DBG(p.dbg) & 4) {
        // dev arm "MFMA only": the kernel's own MFMA count per k-step, operands in registers, no LDS, no DMA, no barrier - the
        // power-limited ceiling of this instruction mix on this part (profiles/r03_gemm_ceiling.json)
        splitx8 fa, fb;
#pragma unroll
        for (int q = 0; q < 8; ++q) { fa[q] = split_hi(0.001f * (float)(lane + q)); fb[q] = split_hi(0.002f * (float)(lane ^ q)); }
        for (int kt = 0; kt < nk; ++kt)
#pragma unroll
            for (int i = 0; i < 2 * TI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = EXCEL_MFMA32(fb, fa, acc[i][j]);
                    acc[i][j] = EXCEL_MFMA32(fa, fb, acc[i][j]);
                    acc[i][j] = EXCEL_MFMA32(fa, fa, acc[i][j]);
                }
    } else if (NSTAGE == 2) {
        issue_tile(0, 0);
        __syncthreads();                  // drains the LDS-DMA (vmcnt(0)) and publishes the tile
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk && !(EXCEL_DBG(p.dbg) & 2);
            if (TI > 4 && PER_WAVE <= 2 * TI && !(EXCEL_DBG(p.dbg) & 1)) {      // (one piece per 16-row tile: 2 TI slots per stage)
                compute(kt & 1, more ? kt + 1 : -1);               // tile kt+1 streams in behind this step's MFMAs
            } else {
                if (more) issue_tile(kt + 1, (kt + 1) & 1);        // in flight during this step's MFMAs
                if (!(EXCEL_DBG(p.dbg) & 1)) compute(kt & 1, -1);
            }
            __syncthreads();
        }
    } else {
        // ring of NSTAGE (3) stages, two k-steps of loads in flight.  Step kt: wait until only the NEWER stage's loads
        // may still be outstanding (counted vmcnt: PER_WAVE per stage), raw barrier (publishes stage kt to all waves AND
        // proves everyone finished reading stage kt-1), refill stage kt-1's slot with tile kt+2, compute stage kt.
        issue_tile(0, 0);
        if (nk > 1) issue_tile(1, 1);
        int stage = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                static_assert(NSTAGE == 2 || PER_WAVE == 6 || PER_WAVE == 8, "add the counted wait for this configuration");
                if (PER_WAVE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk && !(EXCEL_DBG(p.dbg) & 2)) issue_tile(kt + 2, stage == 0 ? NSTAGE - 1 : stage - 1);
            if (!(EXCEL_DBG(p.dbg) & 1)) compute(stage, -1);
            stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
        }
        __syncthreads();
    }

    if (EXCEL_DBG(p.dbg) & 8) {
        // dev arm "no epilogue": the k-loop alone (the accumulators stay live through a store that never happens)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2 * TI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += acc[i][j][e];
        if (sum == 1.2345e-30f) p.C[0] = sum;
        return;
    }
    const bool vec = (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 && (p.hd & 3) == 0;
    if (vec) {
        // every wave is past the last barrier of the k-loop: the staging ring is free -> per-wave transpose scratch
        float* scratch = reinterpret_cast<float*>(smem) + wave * (32 * 68);
        if (p.out_mode == GEMM_OUT_SPLIT_BF16) bf_epilogue_lds<GEMM_OUT_SPLIT_BF16, TI>(p, acc, m0, n0, wm, wn, lane, scratch, nti);
        else if (p.out_mode == GEMM_OUT_QKV_HEADMAJOR) bf_epilogue_lds<GEMM_OUT_QKV_HEADMAJOR, TI>(p, acc, m0, n0, wm, wn, lane, scratch, nti);
        else bf_epilogue_lds<GEMM_OUT_PLAIN, TI>(p, acc, m0, n0, wm, wn, lane, scratch, nti);
    } else {
        if (p.out_mode == GEMM_OUT_SPLIT_BF16) bf_epilogue<GEMM_OUT_SPLIT_BF16, TI>(p, acc, m0, n0, wm, wn, lane, nti);
        else if (p.out_mode == GEMM_OUT_QKV_HEADMAJOR) bf_epilogue<GEMM_OUT_QKV_HEADMAJOR, TI>(p, acc, m0, n0, wm, wn, lane, nti);
        else bf_epilogue<GEMM_OUT_PLAIN, TI>(p, acc, m0, n0, wm, wn, lane, nti);
    }
}

// fp32 [R,K] -> split bf16 [R,2,K]
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ in, u16* __restrict__ out, long long R, int K) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= R * K) return;
    const long long row = i / K;
    const int k = (int)(i % K);
    const f32x4 v = *reinterpret_cast<const f32x4*>(in + i);
    split_t hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = split_hi(v[j]);
        lo[j] = split_hi(v[j] - (float)hi[j]);
    }
    split_t* o = reinterpret_cast<split_t*>(out) + row * 2 * K + split_off(k, 0);
    *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(hi);
    *reinterpret_cast<uint2*>(o + 32) = *reinterpret_cast<const uint2*>(lo);
}

// fp32 [R,K] -> the hi plane alone as a plain 16-bit matrix [R,K] (operand of the two-product GEMM, gemm_w4x2.hip) + the count of
// elements whose lo plane would not be zero (x != (float)hi(x); a NaN counts): 0 <=> the matrix is exactly representable in the split
// type, the precondition of the two-product mode
__global__ __launch_bounds__(256) void pack_hi_kernel(const float* __restrict__ in, u16* __restrict__ out, long long n, unsigned long long* __restrict__ inexact) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    unsigned bad = 0;
    if (i < n) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + i);
        split_t hi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hi[j] = split_hi(v[j]);
            bad += ((float)hi[j] != v[j]) ? 1u : 0u;          // (NaN: true)
        }
        *reinterpret_cast<uint2*>(out + i) = *reinterpret_cast<const uint2*>(hi);
    }
    bad = __builtin_amdgcn_readfirstlane((unsigned)wave_sum((float)bad));      // <= 256 per wave: exact in fp32
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(inexact, (unsigned long long)bad);
}

// V^T in split format vt [B][H*64][2*KP] (row = one (head, d), K axis = token m, zero padded to KP): the K-major B operand of the bf16x3
// A_sum . V GEMM, cut from the split q|k|v the QKV epilogue wrote (rows [hi 64 | lo 64] of v): a pure 16-bit transpose.  Since round 3 only the surgery blocks' A_sum.V GEMM needs it (the row pass reads V
// row-major through the LDS transpose read).
__global__ __launch_bounds__(256) void vt_from_planes_kernel(const u16* __restrict__ qkvs, u16* __restrict__ vt, int H, int N, int KP) {
    // 64 tokens x [hi 64 | lo 64] u16 as 64 dwords per row, pitch 65.  Both phases keep a 2-way LDS bank conflict (SQ_LDS_BANK_CONFLICT /
    // SQ_LDS_IDX_ACTIVE = 0.49 in profiles/r03_pipe_busy.txt - round 3's commit called this layout "conflict-free", which it is not): the
    // 32 lanes of a service group are 2 tokens x 16 chunks in the load phase (bank m + 4 c8: chunks c8 and c8 + 8 collide) and 16 token
    // groups x 2 d-pairs in the transpose phase (bank 4 g4 + dpair: g4 and g4 + 8 collide).  Rounds 1-2 read single u16 down the rows of a
    // 136-u16 pitch (conflict rate 0.82).  Making the transpose reads conflict-free (lanes along dpair) would turn the 128-byte coalesced
    // global stores into 32 scattered 8-byte ones - the worse trade for a 30-us kernel (0.14 ms per step).
    __shared__ unsigned t32[64 * 65];
    const int mt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const u16* src = qkvs + ((((long long)b * 3 + 2) * H + h) * (long long)N) * 128;
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 16; i += 256) {              // 64 tokens x 16 chunks of 8 u16
        const int m = i >> 4, c8 = i & 15;
        const int gm = mt * 64 + m;
        uint4 x = {0u, 0u, 0u, 0u};
        if (gm < N) x = *reinterpret_cast<const uint4*>(src + (long long)gm * 128 + c8 * 8);
        unsigned* d = t32 + m * 65 + c8 * 4;
        d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
    }
    __syncthreads();
    for (int i = tid; i < 2 * 32 * 16; i += 256) {          // thread -> (hi | lo plane, pair of d, 4 consecutive tokens)
        const int half = i >> 9, dpair = (i >> 4) & 31, g4 = i & 15;
        const int m0 = mt * 64 + g4 * 4;
        if (m0 >= KP) continue;
        const unsigned* rp = t32 + (g4 * 4) * 65 + half * 32 + dpair;
        const unsigned w0 = rp[0], w1 = rp[65], w2 = rp[130], w3 = rp[195];
        uint2 even, odd;                                      // d = 2 dpair: the low halves of the four dwords; d + 1: the high halves
        even.x = __builtin_amdgcn_perm(w1, w0, 0x05040100u); even.y = __builtin_amdgcn_perm(w3, w2, 0x05040100u);
        odd.x = __builtin_amdgcn_perm(w1, w0, 0x07060302u);  odd.y = __builtin_amdgcn_perm(w3, w2, 0x07060302u);
        u16* o = vt + (((long long)b * H + h) * 64 + 2 * dpair) * 2 * KP + split_off(m0, 0) + half * 32;
        *reinterpret_cast<uint2*>(o) = even;
        *reinterpret_cast<uint2*>(o + 2 * KP) = odd;
    }
}

int excel_launch_vt_from_planes(const unsigned short* qkvs, unsigned short* vt, int B, int H, int N, int KP, hipStream_t st) {
    ProfScope prof__(PROF_OTHER, st);
    EXCEL_CHECK_ARG((KP % 32) == 0 && KP >= N, "vt_from_planes: KP must be a multiple of 32 and >= N");
    hipLaunchKernelGGL(vt_from_planes_kernel, dim3(cdiv(KP, 64), H, B), dim3(256), 0, st, qkvs, vt, H, N, KP);
    EXCEL_CHECK_LAUNCH("vt_from_planes");
    return EXCEL_OK;
}


int excel_launch_gemm_bf16x3(const GemmBfArgs& p_in, hipStream_t stream) {
    GemmBfArgs p = p_in;
#ifdef EXCEL_DEV
    { static const char* d = getenv("EXCEL_BF_DBG"); if (d) p.dbg = atoi(d); }
#endif
    ProfScope prof__(g_excel_prof_gemm_cat >= 0 ? g_excel_prof_gemm_cat : PROF_GEMM_BF16X3, stream, 2.0 * p.M * (double)p.N * p.K * (p.batch > 1 ? p.batch : 1));
    EXCEL_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && (p.K % TBK) == 0, "gemm_bf16x3: K must be a multiple of %d (K=%d)", TBK, p.K);
    EXCEL_CHECK_ARG(p.out_mode != GEMM_OUT_SPLIT_BF16 || (p.N % 32) == 0, "gemm_bf16x3: split output needs N %% 32 == 0");
    EXCEL_CHECK_ARG((p.lda % 8) == 0 && (p.ldb % 8) == 0 && p.lda >= 2 * p.K && p.ldb >= 2 * p.K, "gemm_bf16x3: bad lda/ldb");
    EXCEL_CHECK_ARG((((uintptr_t)p.A | (uintptr_t)p.B) & 15) == 0, "gemm_bf16x3: operands must be 16-byte aligned");
#ifdef EXCEL_DEV
    static const char* force = getenv("EXCEL_BF_TILE");      // dev knob: "128" | "256x128" | "256" | "320"
    static const bool force_uniform = getenv("EXCEL_BF_UNIFORM") != nullptr;   // dev knob: no mixed-height tiles
#else
    const char* const force = nullptr;
    const bool force_uniform = false;
#endif
    int kind;   // 0: 128x128, 1: 256x128, 2: 256x256, 3: 320x256
    if (force) kind = !strcmp(force, "320") ? 3 : !strcmp(force, "256") ? 2 : (!strcmp(force, "256x128") ? 1 : 0);
    else if (p.M < 2048 || (p.batch > 1)) kind = 0;
    else {
        // The big tiles run one workgroup per CU, so a launch is ceil(tiles / 256) rounds and the last round is mostly
        // idle unless the tile count lands just under a multiple of the CU count (25120 x 768: 594 tiles of 256x128 =
        // 2.3 rounds -> 77 % busy; 237 tiles of 320x256 = 0.93 rounds -> 93 %).  Pick the tile with the best
        // busy fraction x intrinsic efficiency (bytes staged per flop: measured 1.0 / 0.97 / 0.88 / 0.80).
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0; hipDeviceProp_t prop;
            n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        }
        const int bm[4] = {128, 256, 256, 320}, bn[4] = {128, 128, 256, 256}, wg_per_cu[4] = {2, 1, 1, 1};
        const double intrinsic[4] = {0.80, 0.88, 0.97, 1.0};
        double best = -1.0;
        kind = 3;
        for (int k = 0; k < 4; ++k) {
            const long long tiles = (long long)cdiv(p.M, bm[k]) * cdiv(p.N, bn[k]), slots = (long long)n_cu * wg_per_cu[k];
            const long long rounds = (tiles + slots - 1) / slots;
            const double busy = ((double)p.M * p.N) / ((double)rounds * slots * bm[k] * bn[k]);
            if (busy * intrinsic[k] > best) { best = busy * intrinsic[k]; kind = k; }
        }
    }
    const int nb = p.batch > 1 ? p.batch : 1;
#ifdef EXCEL_DEV
    static const bool no_w4 = getenv("EXCEL_BF_W4") && atoi(getenv("EXCEL_BF_W4")) == 0;       // dev knob: the 8-wave kernel for A/B runs
#else
    const bool no_w4 = false;
#endif
    // The four-wave kernel with the hand-placed k-loop (gemm_w4.hip) in its 320- / 256- / 160-row instance, whenever its preconditions
    // hold and its modelled launch time beats the best 8-wave tile's.  8-wave model: algorithmic flops over (tile fill x intrinsic
    // efficiency) x the 320 x 256 tile's measured rate at full fill (345 TFLOP/s fp32-equivalent at K = 768, 400 at K = 3072).
    bool w4_mode_ok = true;
#ifdef EXCEL_DEV
    { static const char* e = getenv("EXCEL_W4_MODES"); if (e) w4_mode_ok = (atoi(e) >> p.out_mode) & 1; }     // dev knob: bit per output mode (plain 1, qkv 2, split 4)
    { static const char* e = getenv("EXCEL_W4_RES"); if (e && atoi(e) == 0 && p.res) w4_mode_ok = false; }      // dev knob: 0 = residual launches stay on the 8-wave kernel
#endif
#ifdef EXCEL_SPLIT_F16
    const bool want_x2 = p.w_lo_zero != 0;          // fp16-valued weights: two-product kernels
#else
    const bool want_x2 = false;                     // (a bf16 hi plane cannot hold an fp16 value: the flag means nothing here)
#endif
    int x2_force = -1;
#ifdef EXCEL_DEV
    { static const char* e = getenv("EXCEL_W4_X2"); if (e) x2_force = atoi(e); }        // dev knob: 0 = three-product kernels, 1 = split-layout weights only
#endif
    const bool x2_on = want_x2 && x2_force != 0;
    if (!no_w4 && w4_mode_ok && nb == 1 && p.M >= 2048) {
        static int n_cu3 = 0;
        if (!n_cu3) {
            int dev = 0; hipDeviceProp_t prop;
            n_cu3 = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        }
        int force_ntm = 0;
#ifdef EXCEL_DEV
        { static const char* e = getenv("EXCEL_W4_NTM"); if (e) force_ntm = atoi(e); }       // dev knob: 10 / 8 / 5, -1 = model only
        if (force && !strcmp(force, "320")) force_ntm = 10;
#endif
        const int bm8[4] = {128, 256, 256, 320}, bn8[4] = {128, 128, 256, 256}, wg8[4] = {2, 1, 1, 1};
        const double in8[4] = {0.80, 0.88, 0.97, 1.0};
        const long long tiles8 = (long long)cdiv(p.M, bm8[kind]) * cdiv(p.N, bn8[kind]), slots8 = (long long)n_cu3 * wg8[kind];
        const double busy8 = ((double)p.M * p.N) / ((double)((tiles8 + slots8 - 1) / slots8) * slots8 * bm8[kind] * bn8[kind]);
        const double kfac = p.K <= 768 ? 0.0 : (p.K >= 3072 ? 1.0 : (p.K - 768) / 2304.0);
        // (advisor, round 5: a forced 8-wave tile - "128" / "256x128" / "256" - must reach the 8-wave kernel: only "320" or no force routes here)
        const bool force8 = force && strcmp(force, "320") != 0;
        double best_us = force ? 1e30 : 2.0 * p.M * (double)p.N * p.K / (busy8 * in8[kind] * (345.0 + 55.0 * kfac) * (x2_on && kind != 3 ? 1.40 : 1.0) * 1e6);
        int best_ntm = 0, best_x2 = 0;
        const int cand[3] = {10, 8, 5};
        for (int c = 0; c < 3 && !force8; ++c) {
            // the compact-weight instance when the plain half matrix is there, else the split-layout one
            const int x2 = !x2_on ? 0 : (x2_force != 1 && excel_gemm_w4_supported(p, cand[c], 2)) ? 2 : 1;
            if (!excel_gemm_w4_supported(p, cand[c], x2)) continue;
            if (force_ntm > 0) { if (cand[c] == force_ntm) { best_ntm = force_ntm; best_x2 = x2; } continue; }
            const double us = excel_gemm_w4_model_us(p, cand[c], n_cu3, x2);
            if (us < best_us) { best_us = us; best_ntm = cand[c]; best_x2 = x2; }
        }
        // a launch of two instances (full rounds of 320-row tiles + the rest in shorter ones) when the model prefers it by more than 1 %
        bool mix_ok = force_ntm == 0 && !force8;
#ifdef EXCEL_DEV
        { static const char* e = getenv("EXCEL_W4_MIX"); if (e && atoi(e) == 0) mix_ok = false; }       // dev knob: uniform launches only
#endif
        if (mix_ok) {
            const int mx2 = !x2_on ? 0 : 2;          // (the split-layout two-product form has no 320-row instance: no two-instance launch)
            int tall = 0, shrt = 0, second = 0;
            const double mus = (x2_on && x2_force == 1) ? 1e30 : excel_gemm_w4_mix_model_us(p, n_cu3, mx2, &tall, &shrt, &second);
            if (mus < 0.99 * best_us) {
#ifdef EXCEL_SPLIT_F16
                if (mx2) return excel_launch_gemm_w4x2_mix(p, tall, shrt, second, stream);
#endif
                return excel_launch_gemm_w4_mix(p, tall, shrt, second, stream);
            }
        }
#ifdef EXCEL_SPLIT_F16
        if (best_ntm && best_x2) return excel_launch_gemm_w4x2(p, best_ntm, best_x2, stream);
#endif
        if (best_ntm) return excel_launch_gemm_w4(p, best_ntm, stream);
    }
    if (kind == 3 && nb == 1 && !force_uniform) {
        // mixed-height row tiles (kernel header): R = rounds of the uniform 320-row tiling; nt = the row tiles that fit into R rounds;
        // `tall` of them must be 320 rows high to cover M, the rest can be 256.  Worth it when the tall tiles leave room in the last round
        // for short ones (tall * tiles_n <= (R - 1) * CUs): then no CU gets R tall tiles.
        static int n_cu2 = 0;
        if (!n_cu2) {
            int dev = 0; hipDeviceProp_t prop;
            n_cu2 = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        }
        const int tiles_n = cdiv(p.N, 256), units = cdiv(p.M, 32);
        const int R = cdiv(cdiv(p.M, 320) * tiles_n, n_cu2);
        const int nt = (R * n_cu2) / tiles_n;
        int tall = (units - 8 * nt + 1) / 2;
        if (tall < 0) tall = 0;
        int shrt = nt - tall;
        while (shrt > 0 && 10 * tall + 8 * (shrt - 1) >= units) --shrt;      // no more row tiles than M needs
        int first = 0;
        bool one_round = false;
#ifdef EXCEL_DEV
        { static const char* e = getenv("EXCEL_BF_FIRST"); if (e) first = atoi(e); }
        { static const char* e = getenv("EXCEL_BF_MIX1"); one_round = e && R == 1; }
#endif
        if ((R >= 2 || one_round) && shrt > 0 && tall <= nt && 10 * tall + 8 * shrt >= units && (one_round || tall * tiles_n <= (R - 1) * n_cu2)) {
            p.mix_tall = tall; p.mix_short = shrt; p.mix_first = first;
            hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 4, 5, 2, true>), dim3((tall + shrt) * tiles_n, 1), dim3(512), 0, stream, p);
            EXCEL_CHECK_LAUNCH("gemm_bf16x3");
            return EXCEL_OK;
        }
    }
#ifdef EXCEL_SPLIT_F16
    if (x2_on && kind != 3) {        // two-product instances of the tiles that small / odd-shaped weight GEMMs land on
        if (kind == 2) hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 4, 4, 2, false, true>), dim3(cdiv(p.M, 256) * cdiv(p.N, 256), nb), dim3(512), 0, stream, p);
        else if (kind == 1) hipLaunchKernelGGL((gemm_bf16x3_kernel<4, 2, 2, 3, false, true>), dim3(cdiv(p.M, 256) * cdiv(p.N, 128), nb), dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 2, 2, 2, false, true>), dim3(cdiv(p.M, 128) * cdiv(p.N, 128), nb), dim3(256), 0, stream, p);
        EXCEL_CHECK_LAUNCH("gemm_f16x2");
        return EXCEL_OK;
    }
#endif
    if (kind == 3) {
        hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 4, 5, 2>), dim3(cdiv(p.M, 320) * cdiv(p.N, 256), nb), dim3(512), 0, stream, p);
    } else if (kind == 2) {
        hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 4, 4, 2>), dim3(cdiv(p.M, 256) * cdiv(p.N, 256), nb), dim3(512), 0, stream, p);
    } else if (kind == 1) {
        hipLaunchKernelGGL((gemm_bf16x3_kernel<4, 2, 2, 3>), dim3(cdiv(p.M, 256) * cdiv(p.N, 128), nb), dim3(512), 0, stream, p);
    } else {
        hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 2, 2, 2>), dim3(cdiv(p.M, 128) * cdiv(p.N, 128), nb), dim3(256), 0, stream, p);
    }
    EXCEL_CHECK_LAUNCH("gemm_bf16x3");
    return EXCEL_OK;
}

int excel_launch_pack_hi(const float* in, void* out, long long R, int K, unsigned long long* inexact, hipStream_t st) {
    ProfScope prof__(PROF_OTHER, st);
    EXCEL_CHECK_ARG(in && out && inexact && R > 0 && K > 0 && ((R * K) % 4) == 0, "pack_hi: bad argument (R * K must be a multiple of 4)");
    hipLaunchKernelGGL(pack_hi_kernel, dim3((unsigned)cdivl(R * K / 4, 256)), dim3(256), 0, st, in, (u16*)out, R * K, inexact);
    EXCEL_CHECK_LAUNCH("pack_hi");
    return EXCEL_OK;
}

int excel_launch_split_bf16(const float* in, void* out, long long R, int K, hipStream_t st) {
    ProfScope prof__(PROF_OTHER, st);
    EXCEL_CHECK_ARG((K % 32) == 0, "split_bf16: K must be a multiple of 32 (blocked hi/lo layout)");
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)cdivl(R * K / 4, 256)), dim3(256), 0, st, in, (u16*)out, R, K);
    EXCEL_CHECK_LAUNCH("split_bf16");
    return EXCEL_OK;
}

}  // namespace EXCEL_SPLIT_NS
