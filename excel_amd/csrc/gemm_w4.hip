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

typedef unsigned short u16;

#ifdef EXCEL_SPLIT_F16
#define W4_MFMA_OP "v_mfma_f32_16x16x32_f16"
#else
#define W4_MFMA_OP "v_mfma_f32_16x16x32_bf16"
#endif

namespace w4 {
constexpr int BN = 256, WTN = 128;
constexpr int ROWB = 128;                           // bytes per staged row
constexpr int B_BYTES = BN * ROWB;                  // 32 768
constexpr int NT_N = WTN / 16;                      // 8 column tiles of 16 per wave
constexpr int B_PIECES = BN / 8 / 4;                // 1-KB DMA pieces of the weight rows per wave and stage: 8
// Geometry of an instance: a wave owns NT_M x 8 accumulator tiles (16 NT_M rows x 128 columns), the workgroup 32 NT_M x 256.
//   NT_M = 10: 320 x 256 (the B = 32 layer shapes: 237 / 711 / 948 tiles), 8: 256 x 256, 5: 160 x 256 (B = 16: 237 tiles for N = 768)
template <int NT_M>
struct Geo {
    static constexpr int BM = 32 * NT_M, WTM = 16 * NT_M;
    static constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int A_PIECES = NT_M, PIECES = A_PIECES + B_PIECES;      // BM / 8 rows per piece / 4 waves = NT_M
    static constexpr int MAIN = NT_M - 2;                 // row tiles in front of the barrier (AGPR accumulators); the last two form the tail (VGPR)
    static constexpr int UNROLL = (NT_M & 1) ? 4 : 2;     // steps until (step * NT_M) % 4 (A-ring phase) and the stage parity repeat
    // DMA pieces of the stage after next that already go out in the TAIL of a step (into the stage its barrier has just freed): only the
    // short instance needs them - its three main row tiles are too few to both issue 13 pieces and cover their latency
    static constexpr int TAILQ = NT_M == 5 ? 8 : 0;
    // pieces per main row tile, at MFMA positions pos(count, k)
    static constexpr int cnt(int i) { return NT_M == 10 ? (i <= 2 ? 4 : i <= 4 ? 3 : 0) : NT_M == 8 ? (i <= 1 ? 5 : i <= 3 ? 3 : 0) : (i == 0 ? 5 : 0); }
    static constexpr int pos(int c, int k) { return c == 5 ? (k == 0 ? 2 : k == 1 ? 6 : k == 2 ? 10 : k == 3 ? 14 : 20) : c == 4 ? (k == 0 ? 2 : k == 1 ? 10 : k == 2 ? 14 : 20) : (k == 0 ? 2 : k == 1 ? 10 : 14); }
    // piece issued behind MFMA n of main row tile i, or -1
    static constexpr int piece_at(int i, int n) {
        int q = TAILQ;
        for (int t = 0; t < i; ++t) q += cnt(t);
        for (int k = 0; k < cnt(i); ++k) if (pos(cnt(i), k) == n) return q + k;
        return -1;
    }
    static_assert(TAILQ + cnt(0) + cnt(1) + cnt(2) + cnt(3) + cnt(4) == PIECES, "every piece has a slot");
};

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

// One launch = cdiv(M, 32 NT_M) x cdiv(N,256) workgroups of 256 threads.  Preconditions (checked by the launcher): K % 64 == 0 (K % 128 for the odd NT_M = 5: the k-loop is unrolled until ring phase and stage parity repeat),
// N, ldc, ldr, hd multiples of 4 (vector epilogue), operand extents below 2^31 bytes (32-bit buffer offsets), batch == 1.
// DBG (development builds only; the shipped library instantiates DBG = 0): timing-only ablation arms, results are wrong -
//   1 no LDS-DMA after the prologue, 2 no fragment reads after the prologue, 4 no barrier in the k-loop, 8 no epilogue, 16 no MFMAs
template <int NT_M, int DBG>
__device__ __forceinline__ void gemm_w4_body(GemmBfArgs p) {
    using namespace w4;
    using G = Geo<NT_M>;
    constexpr int BM = G::BM, WTM = G::WTM, A_BYTES = G::A_BYTES, STAGE_BYTES = G::STAGE_BYTES, A_PIECES = G::A_PIECES, PIECES = G::PIECES;
    constexpr int MAIN = G::MAIN, UNROLL = G::UNROLL, TAILQ = G::TAILQ;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    // (DBG & 128) phase stamps of workgroups 0 / 100 / 200, wave 0: s_memrealtime (100 MHz) at entry, after the prologue's barrier, behind the
    // k-loop and at the end -> `bias` buffer, u64 slots 400 + 8 (blockIdx / 100) + phase
    auto phase_stamp = [&](int ph) {
        if constexpr (DBG & 128) {
            if (wave == 0 && (ph == 0 || ph == 3) && blockIdx.x < 700) {          // every workgroup: entry and end -> slots 500 + 2 b
                const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias))[500 + 2 * blockIdx.x + (ph ? 1 : 0)] = rt;
            }
            if (wave == 0 && (blockIdx.x % 100) == 0 && blockIdx.x < 300) {
                const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias))[400 + 8 * (blockIdx.x / 100) + ph] = rt;
            }
        }
    };
    phase_stamp(0);
#ifdef EXCEL_DEV
    // dev experiment (EXCEL_W4_STAGGER = D us): the first round's workgroups start up to D us apart (8 phases per XCD), so that the
    // epilogue store bursts of the 256 CUs no longer fall on top of each other for the rest of the launch
    if (p.dbg > 0 && blockIdx.x < 256) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                  // 100 MHz
        const unsigned long long ticks = (unsigned long long)((blockIdx.x >> 3) & 7) * (unsigned)p.dbg * 100ull / 8ull;
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    }
#endif
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

    f32x4 accA[MAIN][NT_N];   // AGPR file: row tiles 0 .. MAIN-1
    f32x4 accV[2][NT_N];      // VGPR file: the two tail row tiles
    sfor<0, MAIN>([&](auto I) { sfor<0, NT_N>([&](auto J) { accA[decltype(I)::value][decltype(J)::value] = f32x4{0.f, 0.f, 0.f, 0.f}; }); });
    sfor<0, 2>([&](auto I) { sfor<0, NT_N>([&](auto J) { accV[decltype(I)::value][decltype(J)::value] = f32x4{0.f, 0.f, 0.f, 0.f}; }); });

    splitx8 Ah[4], Al[4];     // ring of A fragments: row tile i of step P (mod UNROLL) sits in slot (P NT_M + i) & 3
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

    // ---- one 32-k step.  P = step index mod UNROLL: data in stage P & 1, A-ring phase P NT_M.  The stage of the NEXT step (k offset
    // `soff_next` bytes) is fetched during this one (and, short instance, its first TAILQ pieces already in the previous step's tail: this
    // step's tail issues those of the step after next, k offset `soff_next2`, into the stage its barrier has just freed).  Past the end of
    // K the offsets are clamped to the last k-block (fetched once more, never read; +1/nk of the L2 -> LDS traffic): the stream has no
    // end-of-loop variant - one copy per step phase, no per-piece branches, no join of differently allocated accumulator sets.
    auto step = [&](auto Pc, int soff_next, int soff_next2) {
        constexpr int P = decltype(Pc)::value, S = P & 1, R = (P * NT_M) & 3, Rn = ((P + 1) * NT_M) & 3;
        // main row tiles, pass-major: lo.hi over the 8 column tiles, then hi.lo, then hi.hi (small terms first)
        sfor<0, MAIN>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            constexpr int slot = (R + i) & 3;
            if constexpr (i > 0) wait_lgkm<2>();              // tile i landed; only tile i + 1's two reads may still be in flight
            sfor<0, 3 * NT_N>([&](auto Nc) {
                constexpr int n = decltype(Nc)::value, pass = n / NT_N, j = n % NT_N;
                if constexpr (i == 0 && pass == 0) wait_lgkm<15 - 2 * j>();        // tail read order: A0, A1, then (Bh, Bl) per column tile
                if constexpr (i == 0 && pass == 1 && j == 0) wait_lgkm<0>();
                if constexpr (pass == 0) mfma_agpr<!(DBG & 16)>(accA[i][j], Al[slot], Bh[j]);
                else if constexpr (pass == 1) mfma_agpr<!(DBG & 16)>(accA[i][j], Ah[slot], Bl[j]);
                else mfma_agpr<!(DBG & 16)>(accA[i][j], Ah[slot], Bh[j]);
                // A fragments two row tiles ahead (ring slot of tile i - 2, retired), behind the first pass
                if constexpr (n == NT_N) read_a(IC<S>{}, IC<i + 2>{}, IC<(R + i + 2) & 3>{});
                // the next stage's DMA pieces, one at a time between MFMAs (Geo::piece_at)
                constexpr int q = G::piece_at(i, n);
                if constexpr (q >= 0 && !(DBG & 1)) dma(IC<q>{}, IC<q + 1>{}, (1 - S) * STAGE_BYTES, soff_next);
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
        // tail: the last two row tiles (VGPR accumulators) column-pair-major; the next step's A tiles 0, 1 first, then each pair's B
        // fragments as soon as its 12 MFMAs have been issued
        read_a(IC<1 - S>{}, IC<0>{}, IC<(Rn + 0) & 3>{});
        read_a(IC<1 - S>{}, IC<1>{}, IC<(Rn + 1) & 3>{});
        constexpr int s8 = (R + NT_M - 2) & 3, s9 = (R + NT_M - 1) & 3;
        sfor<0, NT_N / 2>([&](auto JPc) {
            constexpr int jp = decltype(JPc)::value, j0 = 2 * jp, j1 = j0 + 1;
            mfma_vgpr<!(DBG & 16)>(accV[0][j0], Al[s8], Bh[j0]); mfma_vgpr<!(DBG & 16)>(accV[0][j1], Al[s8], Bh[j1]);
            mfma_vgpr<!(DBG & 16)>(accV[1][j0], Al[s9], Bh[j0]); mfma_vgpr<!(DBG & 16)>(accV[1][j1], Al[s9], Bh[j1]);
            mfma_vgpr<!(DBG & 16)>(accV[0][j0], Ah[s8], Bl[j0]); mfma_vgpr<!(DBG & 16)>(accV[0][j1], Ah[s8], Bl[j1]);
            mfma_vgpr<!(DBG & 16)>(accV[1][j0], Ah[s9], Bl[j0]); mfma_vgpr<!(DBG & 16)>(accV[1][j1], Ah[s9], Bl[j1]);
            mfma_vgpr<!(DBG & 16)>(accV[0][j0], Ah[s8], Bh[j0]); mfma_vgpr<!(DBG & 16)>(accV[0][j1], Ah[s8], Bh[j1]);
            mfma_vgpr<!(DBG & 16)>(accV[1][j0], Ah[s9], Bh[j0]); mfma_vgpr<!(DBG & 16)>(accV[1][j1], Ah[s9], Bh[j1]);
            read_b(IC<1 - S>{}, IC<j0>{});
            read_b(IC<1 - S>{}, IC<j1>{});
            // short instance: two pieces of the step after next, into the stage this step's barrier has freed
            if constexpr (2 * jp + 1 < TAILQ && !(DBG & 1)) dma(IC<2 * jp>{}, IC<2 * jp + 2>{}, S * STAGE_BYTES, soff_next2);
        });
    };

    // ---- prologue: stage 0 <- k-block 0 (and the pieces of k-block 1 a tail would have issued), published; head reads of step 0 in the
    // order the tail issues them
    const int soff_last = (nk - 1) * 128;
    dma(IC<0>{}, IC<PIECES>{}, 0u, 0);
    if constexpr (TAILQ > 0) dma(IC<0>{}, IC<TAILQ>{}, (unsigned)STAGE_BYTES, min(128, soff_last));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    phase_stamp(1);
    read_a(IC<0>{}, IC<0>{}, IC<0>{});
    read_a(IC<0>{}, IC<1>{}, IC<1>{});
    sfor<0, NT_N>([&](auto J) { read_b(IC<0>{}, J); });

    for (int kt = 0; kt < nk; kt += UNROLL) {            // nk is a multiple of UNROLL (a precondition)
        sfor<0, UNROLL>([&](auto Pc) {
            constexpr int P = decltype(Pc)::value;
            step(Pc, min((kt + P + 1) * 128, soff_last), min((kt + P + 2) * 128, soff_last));
        });
    }
    // the tail of the last step read (stale) fragments that nobody uses; retire them and let the matrix pipe drain before the
    // compiler's own accumulator reads (it cannot see that the asm statements above are MFMAs)
    // (the vmcnt(0) also retires the last step's spare fetch: no LDS-DMA may land after this workgroup has ended.  No barrier: the
    // epilogue does not touch LDS)
    // The fragment registers are OPERANDS of this wait: the last step's tail issued ds_reads into them, and the compiler does not know that
    // an asm ds_read completes asynchronously - with the registers dead behind the loop it handed one of them to the epilogue's lane-id
    // computation ABOVE the wait, and the read landed on top of it (a garbage lane id -> wild bias / store addresses; timing dependent:
    // it showed only when another kernel's waves shared the CU, i.e. with the 160- / 256-row instances on two streams).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15"
                 : "+v"(Ah[0]), "+v"(Ah[1]), "+v"(Ah[2]), "+v"(Ah[3]), "+v"(Al[0]), "+v"(Al[1]), "+v"(Al[2]), "+v"(Al[3]),
                   "+v"(Bh[0]), "+v"(Bh[1]), "+v"(Bh[2]), "+v"(Bh[3]), "+v"(Bh[4]), "+v"(Bh[5]), "+v"(Bh[6]), "+v"(Bh[7]),
                   "+v"(Bl[0]), "+v"(Bl[1]), "+v"(Bl[2]), "+v"(Bl[3]), "+v"(Bl[4]), "+v"(Bl[5]), "+v"(Bl[6]), "+v"(Bl[7])
                 :: "memory");
    phase_stamp(2);

    if constexpr (DBG & 8) {        // the accumulators stay live through a store that never happens
        float sum = 0.f;
        sfor<0, MAIN>([&](auto I) { sfor<0, NT_N>([&](auto J) { const f32x4 v = accA[decltype(I)::value][decltype(J)::value]; sum += v[0] + v[1] + v[2] + v[3]; }); });
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
    auto epilogue = [&](auto MODEc, auto RESc, auto FULLc) {
        constexpr int MODE = decltype(MODEc)::value;
        constexpr bool FULL = decltype(FULLc)::value;          // the tile lies inside the matrix (all but the last row / column of tiles): no row / column
                                                               // predicates, no exec juggling around the stores, no clamped addresses
        constexpr bool RES = decltype(RESc)::value;            // residual launches get their own copy (plain output only): the prefetch below is unconditional there
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
        // The residual rows of row tile i+1 are requested as soon as those of tile i have been added - IN FRONT of tile i's stores (one
        // register set).  vmcnt retires in order: residual loads issued behind the previous row tile's stores, where they are used, made
        // every row tile wait for those stores to retire and then for its own loads - ten serial round trips per tile (round 5, read off the
        // disassembly: 8 loads, vmcnt(7..0), 8-16 stores, 8 loads, ...).
        f32x4 rs[NT_N];
        auto load_res = [&](auto I2c) {
            constexpr int i2 = (decltype(I2c)::value + MAIN) % NT_M;
            const int row2 = FULL ? m0 + wm * WTM + 16 * i2 + r16e : min(m0 + wm * WTM + 16 * i2 + r16e, p.M - 1);
            sfor<0, NT_N>([&](auto Jc) {
                constexpr int j = decltype(Jc)::value;
                const int c = FULL ? colw + 32 * (j >> 1) + 4 * (j & 1) : min(colw + 32 * (j >> 1) + 4 * (j & 1), p.N - 4);
                rs[j] = *reinterpret_cast<const f32x4*>(p.res + (long long)row2 * p.ldr + c);
            });
        };
        if constexpr (RES) load_res(IC<0>{});
        sfor<0, NT_M>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            constexpr int i = (I + MAIN) % NT_M;       // the tail row tiles first: their accumulators occupy 64 VGPRs
            const int row = m0 + wm * WTM + 16 * i + r16e;
            const bool row_ok = FULL || row < p.M;
            const int rowc = row_ok ? row : p.M - 1;
            f32x4 v[NT_N];
            sfor<0, NT_N>([&](auto Jc) {
                constexpr int j = decltype(Jc)::value;
                if constexpr (i < MAIN) v[j] = accA[i][j]; else v[j] = accV[i - MAIN][j];
                asm volatile("" : "+v"(v[j]));        // a clean AGPR -> VGPR copy point (left alone, the allocator splits the tiles into
                                                      // 64-bit halves for packed adds and permutes 200 AGPRs at the loop exit)
            });
            if constexpr (RES) {
                sfor<0, NT_N>([&](auto Jc) { constexpr int j = decltype(Jc)::value; v[j] += bias4[j]; });
                if (p.act == GEMM_ACT_QUICKGELU) { quickgelu_tiles<NT_N / 2>(v); quickgelu_tiles<NT_N / 2>(v + NT_N / 2); }    // (eight chains at a time: all sixteen spill one register)
                sfor<0, NT_N>([&](auto Jc) { constexpr int j = decltype(Jc)::value; v[j] += rs[j]; });
                if constexpr (I + 1 < NT_M) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_res(IC<I + 1>{});
                    __builtin_amdgcn_sched_barrier(0);              // ... in front of this row tile's stores
                }
            } else {
                sfor<0, NT_N>([&](auto Jc) { constexpr int j = decltype(Jc)::value; v[j] += bias4[j]; });
                if (p.act == GEMM_ACT_QUICKGELU) { quickgelu_tiles<NT_N / 2>(v); quickgelu_tiles<NT_N / 2>(v + NT_N / 2); }    // (eight chains at a time: all sixteen spill one register)
            }
            if constexpr (MODE == GEMM_OUT_PLAIN) {
                sfor<0, NT_N>([&](auto Jc) {
                    constexpr int j = decltype(Jc)::value;
                    const int c = colw + 32 * (j >> 1) + 4 * (j & 1);
                    if (FULL || (row_ok && c < p.N)) *reinterpret_cast<f32x4*>(p.C + (long long)row * p.ldc + c) = v[j];
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
                    const bool ok0 = (FULL || (row_ok && c < p.N)) && !(DBG & 64), ok1 = (FULL || (row_ok && c + 4 < p.N)) && !(DBG & 64);   // (64: timing arm without stores)
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
    auto run_epilogue = [&](auto FULLc) {
        if (p.out_mode == GEMM_OUT_SPLIT_BF16) epilogue(IC<GEMM_OUT_SPLIT_BF16>{}, std::false_type{}, FULLc);
        else if (p.out_mode == GEMM_OUT_QKV_HEADMAJOR) epilogue(IC<GEMM_OUT_QKV_HEADMAJOR>{}, std::false_type{}, FULLc);
        else if (p.res) epilogue(IC<GEMM_OUT_PLAIN>{}, std::true_type{}, FULLc);
        else epilogue(IC<GEMM_OUT_PLAIN>{}, std::false_type{}, FULLc);
    };
    if (m0 + BM <= p.M && n0 + BN <= p.N) run_epilogue(std::true_type{});       // (wave-uniform)
    else run_epilogue(std::false_type{});
    if constexpr (DBG & 128) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); phase_stamp(3); }
}

// nt_m: 10 (320-row tiles), 8 (256) or 5 (160)
bool excel_gemm_w4_supported(const GemmBfArgs& p, int nt_m) {
    const bool vec = (p.N & 3) == 0 && p.N >= 8 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 && (p.hd & 7) == 0;
    const int kq = 32 * ((nt_m & 1) ? 4 : 2);          // the k-loop is unrolled over 2 (4) steps of 32
    if (p.res && p.out_mode != GEMM_OUT_PLAIN) return false;      // the residual epilogue exists for the plain output only (all the path uses); else the 8-wave kernel
    return (nt_m == 10 || nt_m == 8 || nt_m == 5) && vec && p.batch <= 1 && p.K >= kq && (p.K % kq) == 0 &&
           (long long)p.M * p.lda * 2 < 0x7fffffffLL && (long long)p.N * p.ldb * 2 < 0x7fffffffLL;
}

// Modelled time (us) of one launch of the nt_m instance on n_cu CUs: rounds of tiles (a partial last round counts less) x (prologue + row tiles x (k-steps x 0.233 + epilogue
// 1.8)); calibrated on the B = 32 layer shapes (profiles/r05_w4_arms.txt: a 320-row tile of K = 768 is 56 us of k-loop + 18 of epilogue + 7),
// the short instance pays ~8 % more per row tile for its fragment reads (26 instead of 36 per 240 MFMAs-equivalent).  The launcher compares
// this against the 8-wave tiles' model (gemm_bf16x3.hip).
double excel_gemm_w4_model_us(const GemmBfArgs& p, int nt_m, int n_cu) {
    const long long tiles = (long long)cdiv(p.M, 32 * nt_m) * cdiv(p.N, w4::BN);
    const long long full = tiles / n_cu, rem = tiles - full * n_cu;
    const double per_row_tile = (p.K / 32) * 0.233 * (nt_m == 5 ? 1.08 : nt_m == 8 ? 1.02 : 1.0) + 1.8;
    // a partly filled last round is cheaper than a full one (fewer CUs share the power budget and the fabric): 0.45 + 0.55 x fill of a full
    // round's time, fitted on the B = 16 shapes (profiles/r05b_b16_shapes.txt: 360 tiles 140.6 us, 480 tiles 163.1, 624 tiles 223.8)
    const double last = rem ? 0.45 + 0.55 * (double)rem / n_cu : 0.0;
    return ((double)full + last) * (7.0 + nt_m * per_row_tile);
}

// one plain __global__ function per instance (a kernel TEMPLATE launched from inside a function template lost its host-side stub)
#define W4_KERNEL(NT, DBG) __global__ __launch_bounds__(256, 1) void gemm_w4_kernel_##NT##_##DBG(GemmBfArgs p) { gemm_w4_body<NT, DBG>(p); }
W4_KERNEL(10, 0) W4_KERNEL(8, 0) W4_KERNEL(5, 0)
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

int excel_launch_gemm_w4(const GemmBfArgs& p, int nt_m, hipStream_t stream) {
    EXCEL_CHECK_ARG(excel_gemm_w4_supported(p, nt_m), "gemm_w4: unsupported problem (vector epilogue, batch 1, K %% 64 (128) == 0, operands below 2 GB)");
    launch_w4(p, nt_m, stream);
    EXCEL_CHECK_LAUNCH("gemm_w4");
    return EXCEL_OK;
}

}  // namespace EXCEL_SPLIT_NS
