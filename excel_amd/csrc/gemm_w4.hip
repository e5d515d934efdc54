// "w4" bf16x3 GEMM: the 320 x 256 tile of gemm_bf16x3.hip on FOUR waves, one per SIMD, with a hand-placed instruction stream.
//
//   C[M,N] (fp32 or split) = act(A_split[M,K] . W_split[N,K]^T + bias) + residual          (same operands, layouts and epilogues)
//
// Why a second kernel (rounds 3 and 4 measured this): the 8-wave kernel's k-loop takes 1.49x its own MFMA stream.  Its
// waves own 160 x 64 outputs - 28 fragment reads per 120 MFMAs - and all eight of them stop at a __syncthreads() every 32 k, read
// their B fragments and only then restart the matrix pipe.  Here a wave owns 160 x 128:
//   * 36 ds_read_b128 per 240 MFMAs (0.15 instead of 0.23 LDS fragment bytes per MFMA: under the socket power cap every LDS byte
//     is clock), 320 accumulator registers = 256 AGPRs (row tiles 0-7) + 64 VGPRs (row tiles 8, 9) of the 512-register budget of a
//     one-wave-per-SIMD kernel;
//   * the compiler only allocates registers: every instruction of the k-loop is an `asm volatile` statement (MFMA with an explicit
//     accumulator register class, ds_read_b128 with immediate offsets, counted s_waitcnt) or an LDS-DMA builtin between them, in
//     program order - hipcc would otherwise shuttle accumulator tiles between the two register files (340 v_accvgpr moves per
//     k-step, round 3) and drain the DMA queue in front of every LDS read it can see;
//   * ONE barrier per 32-k step, placed after row tile 7 of 10, and nothing waits behind it: by then every A fragment of the step
//     has been read (tiles 8 and 9 are fetched two tiles ahead into a 4-deep register ring), so the barrier both publishes the next
//     stage (each wave waited for its own DMA pieces) and frees the current one.  The last two row tiles run column-pair-major, so
//     the B fragments of a column pair are dead after 12 MFMAs and are re-loaded from the NEXT stage right there; the next step's
//     first two A tiles are fetched at the head of this tail.  The matrix pipe never waits for an LDS round trip behind a barrier;
//   * MFMAs on one accumulator are 8 apart (pass-major over the 8 column tiles of a row tile; 4 apart in the tail): a single wave
//     has no partner to fill a dependent-accumulator wait (the first attempt, round 3: +44 % on the bare MFMA stream);
//   * the 18 LDS-DMA pieces of the next stage go out during row tiles 0-4 (4,4,4,3,3), one between MFMAs, through a buffer
//     descriptor: 18 loop-invariant 32-bit lane offsets + one scalar k offset (64-bit lane pointers spilled in round 3).
// Stage layout, swizzle and fragment addressing are the 8-wave kernel's: row = [hi 32 | lo 32] bf16 = 128 B = 8 chunks of 16 B,
// chunk c stored at slot c ^ ((row >> 1) & 7); A rows 0..319, then B rows 0..255; two stages = 147 456 B.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "excel_internal.h"

namespace EXCEL_SPLIT_NS {

typedef unsigned short u16;

#ifdef EXCEL_SPLIT_F16
#define W4_MFMA_OP "v_mfma_f32_16x16x32_f16"
#else
#define W4_MFMA_OP "v_mfma_f32_16x16x32_bf16"
#endif

namespace w4 {
constexpr int BM = 320, BN = 256, WTM = 160, WTN = 128;
constexpr int ROWB = 128;                           // bytes per staged row
constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;   // 40 960 + 32 768 = 73 728
constexpr int NT_M = WTM / 16, NT_N = WTN / 16;      // 10 x 8 accumulator tiles of 16 x 16 per wave
constexpr int A_PIECES = BM / 8 / 4, B_PIECES = BN / 8 / 4, PIECES = A_PIECES + B_PIECES;   // 1-KB DMA pieces per wave and stage: 10 + 8

template <int I> using IC = std::integral_constant<int, I>;
template <int B, int E, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B < E) {
        f(IC<B>{});
        sfor<B + 1, E>(f);
    }
}

// accumulate-in-place MFMA with the accumulator in the AGPR file (row tiles 0-7) or the VGPR file (row tiles 8, 9).  `a` = activation
// fragment, `b` = weight fragment; the instruction gets them SWAPPED (srcA = weights): the tile comes out transposed, a lane then holds
// consecutive COLUMNS of one row (epilogue)
template <bool ON = true>
__device__ __forceinline__ void mfma_agpr(f32x4& c, const splitx8& a, const splitx8& b) {
    if constexpr (!ON) { asm volatile("" : "+a"(c) : "v"(a), "v"(b)); return; }
    asm volatile(W4_MFMA_OP " %0, %2, %1, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <bool ON = true>
__device__ __forceinline__ void mfma_vgpr(f32x4& c, const splitx8& a, const splitx8& b) {
    if constexpr (!ON) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return; }
    asm volatile(W4_MFMA_OP " %0, %2, %1, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int OFF>
__device__ __forceinline__ void lds_rd(splitx8& dst, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read immediate offset");
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    dst = __builtin_bit_cast(splitx8, v);
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
}
}  // namespace w4

// One launch = cdiv(M,320) x cdiv(N,256) workgroups of 256 threads.  Preconditions (checked by the launcher): K % 64 == 0 (an even number of 32-k steps),
// N, ldc, ldr, hd multiples of 4 (vector epilogue), operand extents below 2^31 bytes (32-bit buffer offsets), batch == 1.
// DBG (development builds only; the shipped library instantiates DBG = 0): timing-only ablation arms, results are wrong -
//   1 no LDS-DMA after the prologue, 2 no fragment reads after the prologue, 4 no barrier in the k-loop, 8 no epilogue, 16 no MFMAs
template <int DBG>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(GemmBfArgs p) {
    using namespace w4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = id / tiles_n, tn = id - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA: piece q of this wave covers staged rows seg * 8 .. + 7 with seg = wave + 4 q (A: q < 10, B: q - 10 < 8).  The LDS
    // image is lane-linear (lane l -> row l >> 3, slot l & 7), so the swizzle goes on the SOURCE chunk: c = slot ^ ((row >> 1) & 7).
    typedef __attribute__((address_space(3))) unsigned char* lds_bptr;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7fffffff, 0x00020000);
    int voff[PIECES];
    sfor<0, PIECES>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        constexpr bool isB = q >= A_PIECES;
        const int seg = wave + 4 * (isB ? q - A_PIECES : q);
        const int row_l = seg * 8 + (lane >> 3);
        // A rows: slot = chunk ^ ((row >> 1) & 7); B rows: slot = chunk ^ swz_b(row) - the weight fragments are read with a permuted
        // row <-> lane map (epilogue), and this is the swizzle that keeps THOSE reads conflict-free
        const int c = (lane & 7) ^ (isB ? (((row_l >> 1) & 1) + 2 * ((row_l >> 3) & 3)) : ((row_l >> 1) & 7));
        const int grow = isB ? min(n0 + row_l, p.N - 1) : min(m0 + row_l, p.M - 1);     // rows past the edge re-read the last row (never stored)
        voff[q] = grow * (isB ? p.ldb : p.lda) * 2 + c * 16;
    });
    const unsigned smem_base = lds_addr(smem);
    const unsigned dma_base = __builtin_amdgcn_readfirstlane(smem_base + wave * 1024);   // + stage * STAGE_BYTES + (B: A_BYTES) + q' * 4096
    // pieces [Q0, Q1) of the stage at byte offset `stage_off`, k-block offset `soff` bytes into every row
    auto dma = [&](auto Q0c, auto Q1c, unsigned stage_off, int soff) {
        constexpr int Q0 = decltype(Q0c)::value, Q1 = decltype(Q1c)::value;
        unsigned dst = dma_base + stage_off;
        asm volatile("" : "+s"(dst));             // opaque: keeps the m0 values one s_add each instead of 18 hoisted (and spilled) SGPRs
        sfor<Q0, Q1>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            if constexpr (q < A_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_bptr)(unsigned long long)(dst + q * 4096), 16, voff[q], soff, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_bptr)(unsigned long long)(dst + A_BYTES + (q - A_PIECES) * 4096), 16, voff[q], soff, 0, 0);
        });
    };

    // ---- fragment addresses (bytes): row tile i of this wave's A rows = a_* + i * 2048, column tile j = b_* + j * 2048; stage 1 = + STAGE_BYTES
    const int r16 = lane & 15, kg = lane >> 4, sw = (r16 >> 1) & 7;
    const unsigned a_hi = smem_base + (wm * WTM + r16) * ROWB + ((kg ^ sw) * 16);
    const unsigned a_lo = smem_base + (wm * WTM + r16) * ROWB + (((4 + kg) ^ sw) * 16);
    // weight fragment of column tile j = 2 jp + t: lane r reads staged row 32 jp + 4 t + brow, brow = 8 (r / 4) + (r % 4); swz_b(brow) does
    // not depend on jp, t
    const int brow = 8 * (r16 >> 2) + (r16 & 3), swb = ((brow >> 1) & 1) + 2 * ((brow >> 3) & 3);
    const unsigned b_hi = smem_base + A_BYTES + (wn * WTN + brow) * ROWB + ((kg ^ swb) * 16);
    const unsigned b_lo = smem_base + A_BYTES + (wn * WTN + brow) * ROWB + (((4 + kg) ^ swb) * 16);
    const unsigned a_hi1 = a_hi + STAGE_BYTES, a_lo1 = a_lo + STAGE_BYTES, b_hi1 = b_hi + STAGE_BYTES, b_lo1 = b_lo + STAGE_BYTES;

    f32x4 accA[8][NT_N];      // AGPR file
    f32x4 accV[2][NT_N];      // VGPR file
    sfor<0, 8>([&](auto I) { sfor<0, NT_N>([&](auto J) { accA[decltype(I)::value][decltype(J)::value] = f32x4{0.f, 0.f, 0.f, 0.f}; }); });
    sfor<0, 2>([&](auto I) { sfor<0, NT_N>([&](auto J) { accV[decltype(I)::value][decltype(J)::value] = f32x4{0.f, 0.f, 0.f, 0.f}; }); });

    splitx8 Ah[4], Al[4];     // ring of A fragments: row tile i of a step with parity P sits in slot (2 P + i) & 3
    splitx8 Bh[NT_N], Bl[NT_N];

    const int nk = p.K / 32;
    int stamp_idx = 0;            // (DBG & 128 only)

    // reads of the head of a step from the stage at `S` (compile-time 0 / 1): A tiles 0 and 1, then (tail only) the B column pairs
    auto read_a = [&](auto Sc, auto Ic, auto SLOTc) {
        constexpr int S = decltype(Sc)::value, i = decltype(Ic)::value, slot = decltype(SLOTc)::value;
        if constexpr (DBG & 2) return;
        lds_rd<i * 2048>(Ah[slot], S ? a_hi1 : a_hi);
        lds_rd<i * 2048>(Al[slot], S ? a_lo1 : a_lo);
    };
    auto read_b = [&](auto Sc, auto Jc) {
        constexpr int S = decltype(Sc)::value, j = decltype(Jc)::value;
        if constexpr (DBG & 2) return;
        lds_rd<(32 * (j >> 1) + 4 * (j & 1)) * ROWB>(Bh[j], S ? b_hi1 : b_hi);
        lds_rd<(32 * (j >> 1) + 4 * (j & 1)) * ROWB>(Bl[j], S ? b_lo1 : b_lo);
    };

    // ---- one 32-k step.  P = step parity: data in stage P, A-ring phase 2 P.  The stage of the NEXT step (k offset `soff_next` bytes) is
    // fetched during this one.  The last step fetches its own k-block once more into the other stage (nobody reads it; +1/nk of the
    // L2 -> LDS traffic): the stream then has no end-of-loop variant - one copy of each parity, no per-piece branches, and no join of
    // differently allocated accumulator sets (a peeled last pair made the compiler permute 256 AGPRs and spill 24 VGPRs at the seam).
    auto step = [&](auto Pc, int soff_next) {
        constexpr int P = decltype(Pc)::value;
        // row tiles 0..7, pass-major: lo.hi over the 8 column tiles, then hi.lo, then hi.hi (small terms first)
        sfor<0, 8>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            constexpr int slot = (2 * P + i) & 3;
            if constexpr (i > 0) wait_lgkm<2>();              // tile i landed; only tile i + 1's two reads may still be in flight
            sfor<0, 3 * NT_N>([&](auto Nc) {
                constexpr int n = decltype(Nc)::value, pass = n / NT_N, j = n % NT_N;
                if constexpr (i == 0 && pass == 0) wait_lgkm<15 - 2 * j>();        // tail read order: A0, A1, then (Bh, Bl) per column tile
                if constexpr (i == 0 && pass == 1 && j == 0) wait_lgkm<0>();
                if constexpr (pass == 0) mfma_agpr<!(DBG & 16)>(accA[i][j], Al[slot], Bh[j]);
                else if constexpr (pass == 1) mfma_agpr<!(DBG & 16)>(accA[i][j], Ah[slot], Bl[j]);
                else mfma_agpr<!(DBG & 16)>(accA[i][j], Ah[slot], Bh[j]);
                // A fragments two row tiles ahead (ring slot of tile i - 2, retired), behind the first pass
                if constexpr (n == NT_N) read_a(IC<P>{}, IC<i + 2>{}, IC<(2 * P + i + 2) & 3>{});
                // the next stage's DMA pieces: 4,4,4,3,3 over row tiles 0..4, one piece between MFMAs
                if constexpr (i <= 4 && (n == 2 || n == 10 || n == 14 || (n == 20 && i <= 2))) {
                    constexpr int q = (i <= 2 ? 4 * i : 12 + 3 * (i - 3)) + (n == 2 ? 0 : n == 10 ? 1 : n == 14 ? 2 : 3);
                    if constexpr (!(DBG & 1)) dma(IC<q>{}, IC<q + 1>{}, (1 - P) * STAGE_BYTES, soff_next);
                }
            });
        });
        // every read of stage P has been issued; this wave's pieces of stage 1 - P have landed -> one barrier publishes and frees
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (DBG & 128) {        // cycle stamps (development): s_memtime before / after the barrier, wave 0 of workgroup 0 -> the `bias` buffer
            if (blockIdx.x == 0 && wave == 0) {
                unsigned long long t0 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                unsigned long long rt = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz: shader clock = d(memtime) / d(memrealtime) x 100 MHz
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias))[2 * stamp_idx] = t0;
                if (lane == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias))[256 + stamp_idx] = rt;
            }
        }
        if constexpr (!(DBG & 4)) __builtin_amdgcn_s_barrier();
        if constexpr (DBG & 128) {
            if (blockIdx.x == 0 && wave == 0) {
                unsigned long long t1 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias))[2 * stamp_idx + 1] = t1;
                ++stamp_idx;
            }
        }
        // tail: row tiles 8, 9 (VGPR accumulators) column-pair-major; the next step's A tiles 0, 1 first, then each pair's B fragments
        // as soon as its 12 MFMAs have been issued
        read_a(IC<1 - P>{}, IC<0>{}, IC<(2 * (1 - P) + 0) & 3>{});
        read_a(IC<1 - P>{}, IC<1>{}, IC<(2 * (1 - P) + 1) & 3>{});
        constexpr int s8 = (2 * P + 8) & 3, s9 = (2 * P + 9) & 3;
        sfor<0, NT_N / 2>([&](auto JPc) {
            constexpr int j0 = 2 * decltype(JPc)::value, j1 = j0 + 1;
            mfma_vgpr<!(DBG & 16)>(accV[0][j0], Al[s8], Bh[j0]); mfma_vgpr<!(DBG & 16)>(accV[0][j1], Al[s8], Bh[j1]);
            mfma_vgpr<!(DBG & 16)>(accV[1][j0], Al[s9], Bh[j0]); mfma_vgpr<!(DBG & 16)>(accV[1][j1], Al[s9], Bh[j1]);
            mfma_vgpr<!(DBG & 16)>(accV[0][j0], Ah[s8], Bl[j0]); mfma_vgpr<!(DBG & 16)>(accV[0][j1], Ah[s8], Bl[j1]);
            mfma_vgpr<!(DBG & 16)>(accV[1][j0], Ah[s9], Bl[j0]); mfma_vgpr<!(DBG & 16)>(accV[1][j1], Ah[s9], Bl[j1]);
            mfma_vgpr<!(DBG & 16)>(accV[0][j0], Ah[s8], Bh[j0]); mfma_vgpr<!(DBG & 16)>(accV[0][j1], Ah[s8], Bh[j1]);
            mfma_vgpr<!(DBG & 16)>(accV[1][j0], Ah[s9], Bh[j0]); mfma_vgpr<!(DBG & 16)>(accV[1][j1], Ah[s9], Bh[j1]);
            read_b(IC<1 - P>{}, IC<j0>{});
            read_b(IC<1 - P>{}, IC<j1>{});
        });
    };

    // ---- prologue: stage 0 <- k-block 0, published; head reads of step 0 in the order the tail issues them
    dma(IC<0>{}, IC<PIECES>{}, 0u, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_a(IC<0>{}, IC<0>{}, IC<0>{});
    read_a(IC<0>{}, IC<1>{}, IC<1>{});
    sfor<0, NT_N>([&](auto J) { read_b(IC<0>{}, J); });

    const int soff_last = (nk - 1) * 128;
    for (int kt = 0; kt < nk; kt += 2) {                 // nk is even (K % 64 == 0 is a precondition)
        step(IC<0>{}, (kt + 1) * 128);
        step(IC<1>{}, min((kt + 2) * 128, soff_last));
    }
    // the tail of the last step read (stale) fragments that nobody uses; retire them and let the matrix pipe drain before the
    // compiler's own accumulator reads (it cannot see that the asm statements above are MFMAs)
    // (the vmcnt(0) also retires the last step's spare fetch: no LDS-DMA may land after this workgroup has ended.  No barrier: the
    // epilogue does not touch LDS)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");

    if constexpr (DBG & 8) {        // the accumulators stay live through a store that never happens
        float sum = 0.f;
        sfor<0, 8>([&](auto I) { sfor<0, NT_N>([&](auto J) { const f32x4 v = accA[decltype(I)::value][decltype(J)::value]; sum += v[0] + v[1] + v[2] + v[3]; }); });
        sfor<0, 2>([&](auto I) { sfor<0, NT_N>([&](auto J) { const f32x4 v = accV[decltype(I)::value][decltype(J)::value]; sum += v[0] + v[1] + v[2] + v[3]; }); });
        if (sum == 1.2345e-30f) p.C[0] = sum;
        return;
    }
    // ---- epilogue, straight from the registers (no LDS transpose).  The MFMAs were issued with the operand roles swapped
    // (srcA = weight fragment, srcB = activation fragment), so an accumulator tile holds C TRANSPOSED: lane (r = lane % 16, g = lane / 16),
    // register e = C[row 16 i + r][column c(j, g, e)].  The weight fragment of column tile j = 2 jp + t reads the staged weight rows
    // 32 jp + 8 (r / 4) + 4 t + (r % 4), which makes c = 32 jp + 8 g + 4 t + e: a lane owns EIGHT consecutive columns per tile pair -
    // fp32 rows go out as 2 x 16 B per lane (128 contiguous bytes per row and instruction pair), split rows as one 16-byte hi and one
    // 16-byte lo store (a whole 64-byte hi / lo segment per row), bias and residual come in the same shape.  (The transposing epilogue
    // of the 8-wave kernel, on one wave per SIMD with nothing to hide its LDS round trips, was 25 % of this kernel: 43 k of 174 k cycles
    // per tile, profiles/r05_w4_arms.txt.)
    int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // re-derived behind the loop: values computed from
    asm volatile("" : "+v"(lane_e));                                                    // `lane` up front would be carried through it (spills)
    const int g4 = lane_e >> 4, r16e = lane_e & 15;
    const int colw = n0 + wn * WTN + 8 * g4;                    // + 32 jp: first of this lane's 8 columns in pair jp
    f32x4 bias4[NT_N];
    sfor<0, NT_N>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        const int c = colw + 32 * (j >> 1) + 4 * (j & 1);
        bias4[j] = (p.bias && c < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    });
    // one specialised copy per output mode (a run-time mode inside the row loop keeps the store addresses of all three modes live at once
    // and spills)
    auto epilogue = [&](auto MODEc) {
        constexpr int MODE = decltype(MODEc)::value;
        // head-major q|k|v scatter: the 8 columns of a pair stay inside one head (hd % 8 == 0): per pair the offset of (type, head, d)
        int qoff[NT_N / 2];                                          // (row indices of the [B,3,H,tokN] planes: < 2^31 rows)
        int qdd[NT_N / 2];
        if constexpr (MODE == GEMM_OUT_QKV_HEADMAJOR) {
            const int D = p.heads * p.hd;
            sfor<0, NT_N / 2>([&](auto JPc) {
                constexpr int jp = decltype(JPc)::value;
                const int c = min(colw + 32 * jp, p.N - 8);
                const int qt = c / D, rem = c - qt * D, qh = rem / p.hd;
                qdd[jp] = rem - qh * p.hd;
                qoff[jp] = (qt * p.heads + qh) * p.tokN;      // + b * 3 * heads * tokN + n  -> row index of the [.., tokN, hd] planes
            });
        }
        sfor<0, NT_M>([&](auto Ic) {
            constexpr int i = (decltype(Ic)::value + 8) % NT_M;          // row tiles 8, 9 first: their accumulators occupy 64 VGPRs
            const int row = m0 + wm * WTM + 16 * i + r16e;
            const bool row_ok = row < p.M;
            const int rowc = row_ok ? row : p.M - 1;
            f32x4 v[NT_N];
            sfor<0, NT_N>([&](auto Jc) {
                constexpr int j = decltype(Jc)::value;
                if constexpr (i < 8) v[j] = accA[i][j]; else v[j] = accV[i - 8][j];
                asm volatile("" : "+v"(v[j]));        // a clean AGPR -> VGPR copy point (left alone, the allocator splits the tiles into
                                                      // 64-bit halves for packed adds and permutes 200 AGPRs at the loop exit)
            });
            if (p.res) {
                f32x4 rs[NT_N];
                sfor<0, NT_N>([&](auto Jc) {
                    constexpr int j = decltype(Jc)::value;
                    const int c = min(colw + 32 * (j >> 1) + 4 * (j & 1), p.N - 4);
                    rs[j] = *reinterpret_cast<const f32x4*>(p.res + (long long)rowc * p.ldr + c);
                });
                sfor<0, NT_N>([&](auto Jc) { constexpr int j = decltype(Jc)::value; v[j] += bias4[j]; });
                if (p.act == GEMM_ACT_QUICKGELU)
                    sfor<0, NT_N>([&](auto Jc) {
                        constexpr int j = decltype(Jc)::value;
    #pragma unroll
                        for (int q = 0; q < 4; ++q) v[j][q] = v[j][q] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[j][q]));
                    });
                sfor<0, NT_N>([&](auto Jc) { constexpr int j = decltype(Jc)::value; v[j] += rs[j]; });
            } else {
                sfor<0, NT_N>([&](auto Jc) { constexpr int j = decltype(Jc)::value; v[j] += bias4[j]; });
                if (p.act == GEMM_ACT_QUICKGELU)
                    sfor<0, NT_N>([&](auto Jc) {
                        constexpr int j = decltype(Jc)::value;
    #pragma unroll
                        for (int q = 0; q < 4; ++q) v[j][q] = v[j][q] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[j][q]));
                    });
            }
            if constexpr (MODE == GEMM_OUT_PLAIN) {
                sfor<0, NT_N>([&](auto Jc) {
                    constexpr int j = decltype(Jc)::value;
                    const int c = colw + 32 * (j >> 1) + 4 * (j & 1);
                    if (row_ok && c < p.N) *reinterpret_cast<f32x4*>(p.C + (long long)row * p.ldc + c) = v[j];
                });
            } else {
                int qrow = 0;
                if constexpr (MODE == GEMM_OUT_QKV_HEADMAJOR) {
                    const int b = rowc / p.tokN, n = rowc - b * p.tokN;
                    qrow = b * 3 * p.heads * p.tokN + n;
                }
                sfor<0, NT_N / 2>([&](auto JPc) {
                    constexpr int jp = decltype(JPc)::value;
                    const int c = colw + 32 * jp;
                    const bool ok0 = row_ok && c < p.N && !(DBG & 64), ok1 = row_ok && c + 4 < p.N && !(DBG & 64);   // (64: timing arm without stores)
                    unsigned hi[4], lo[4];          // 8 columns: packed pairs, already in store order
    #pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        split_pair(v[2 * jp][2 * q], v[2 * jp][2 * q + 1], hi[q], lo[q]);
                        split_pair(v[2 * jp + 1][2 * q], v[2 * jp + 1][2 * q + 1], hi[2 + q], lo[2 + q]);
                    }
                    if constexpr (DBG & 32) {       // timing arm: no split arithmetic (the raw bits of four of the values)
                        *reinterpret_cast<f32x4*>(hi) = v[2 * jp];
                        *reinterpret_cast<f32x4*>(lo) = v[2 * jp + 1];
                    }
                    if constexpr (MODE == GEMM_OUT_SPLIT_BF16) {
                        split_t* o = reinterpret_cast<split_t*>(p.Cs) + (long long)row * 2 * p.N + split_off(c, 0);
                        if (ok1) {
                            *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(hi);
                            *reinterpret_cast<uint4*>(o + 32) = *reinterpret_cast<const uint4*>(lo);
                        } else if (ok0) {
                            *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(hi);
                            *reinterpret_cast<uint2*>(o + 32) = *reinterpret_cast<const uint2*>(lo);
                        }
                    } else {   // q|k|v head-major: fp32 (exact-mode consumers) or [hi hd | lo hd] planes for the bf16x3 attention
                        const long long rowidx = (long long)(qoff[jp] + qrow);
                        if (!p.qkv_split) {
                            float* o = p.C + rowidx * p.hd + qdd[jp];
                            if (ok0) *reinterpret_cast<f32x4*>(o) = v[2 * jp];
                            if (ok1) *reinterpret_cast<f32x4*>(o + 4) = v[2 * jp + 1];
                        } else {
                            split_t* o = reinterpret_cast<split_t*>(p.qkv_split) + rowidx * 2 * p.hd + qdd[jp];
                            if (ok1) {
                                *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(hi);
                                *reinterpret_cast<uint4*>(o + p.hd) = *reinterpret_cast<const uint4*>(lo);
                            } else if (ok0) {
                                *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(hi);
                                *reinterpret_cast<uint2*>(o + p.hd) = *reinterpret_cast<const uint2*>(lo);
                            }
                        }
                    }
                });
            }
            __builtin_amdgcn_sched_barrier(0);      // one row tile at a time: hoisting the next tiles' accumulator reads and residual loads spills
        });
    };
    if (p.out_mode == GEMM_OUT_SPLIT_BF16) epilogue(IC<GEMM_OUT_SPLIT_BF16>{});
    else if (p.out_mode == GEMM_OUT_QKV_HEADMAJOR) epilogue(IC<GEMM_OUT_QKV_HEADMAJOR>{});
    else epilogue(IC<GEMM_OUT_PLAIN>{});
}

bool excel_gemm_w4_supported(const GemmBfArgs& p) {
    const bool vec = (p.N & 3) == 0 && p.N >= 8 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 && (p.hd & 7) == 0;
    return vec && p.batch <= 1 && p.K >= 64 && (p.K % 64) == 0 && (long long)p.M * p.lda * 2 < 0x7fffffffLL && (long long)p.N * p.ldb * 2 < 0x7fffffffLL;
}

int excel_launch_gemm_w4(const GemmBfArgs& p, hipStream_t stream) {
    EXCEL_CHECK_ARG(excel_gemm_w4_supported(p), "gemm_w4: unsupported problem (vector epilogue, batch 1, K %% 64 == 0, operands below 2 GB)");
    const dim3 grid(cdiv(p.M, w4::BM) * cdiv(p.N, w4::BN));
#ifdef EXCEL_DEV
    static const int dbg = getenv("EXCEL_W4_DBG") ? atoi(getenv("EXCEL_W4_DBG")) : 0;
    switch (dbg) {
        case 1: hipLaunchKernelGGL(gemm_w4_kernel<1>, grid, dim3(256), 0, stream, p); break;
        case 2: hipLaunchKernelGGL(gemm_w4_kernel<2>, grid, dim3(256), 0, stream, p); break;
        case 3: hipLaunchKernelGGL(gemm_w4_kernel<3>, grid, dim3(256), 0, stream, p); break;
        case 4: hipLaunchKernelGGL(gemm_w4_kernel<4>, grid, dim3(256), 0, stream, p); break;
        case 7: hipLaunchKernelGGL(gemm_w4_kernel<7>, grid, dim3(256), 0, stream, p); break;
        case 8: hipLaunchKernelGGL(gemm_w4_kernel<8>, grid, dim3(256), 0, stream, p); break;
        case 9: hipLaunchKernelGGL(gemm_w4_kernel<9>, grid, dim3(256), 0, stream, p); break;
        case 10: hipLaunchKernelGGL(gemm_w4_kernel<10>, grid, dim3(256), 0, stream, p); break;
        case 15: hipLaunchKernelGGL(gemm_w4_kernel<15>, grid, dim3(256), 0, stream, p); break;
        case 24: hipLaunchKernelGGL(gemm_w4_kernel<24>, grid, dim3(256), 0, stream, p); break;
        case 32: hipLaunchKernelGGL(gemm_w4_kernel<32>, grid, dim3(256), 0, stream, p); break;
        case 64: hipLaunchKernelGGL(gemm_w4_kernel<64>, grid, dim3(256), 0, stream, p); break;
        case 136: hipLaunchKernelGGL(gemm_w4_kernel<136>, grid, dim3(256), 0, stream, p); break;
        case 143: hipLaunchKernelGGL(gemm_w4_kernel<143>, grid, dim3(256), 0, stream, p); break;
        case 96: hipLaunchKernelGGL(gemm_w4_kernel<96>, grid, dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL(gemm_w4_kernel<0>, grid, dim3(256), 0, stream, p);
    }
#else
    hipLaunchKernelGGL(gemm_w4_kernel<0>, grid, dim3(256), 0, stream, p);
#endif
    EXCEL_CHECK_LAUNCH("gemm_w4");
    return EXCEL_OK;
}

}  // namespace EXCEL_SPLIT_NS
