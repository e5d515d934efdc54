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
constexpr int EPI_PITCH = 132;                       // floats per row of the epilogue's transpose scratch (32 rows x 128 columns per wave)

template <int I> using IC = std::integral_constant<int, I>;
template <int B, int E, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B < E) {
        f(IC<B>{});
        sfor<B + 1, E>(f);
    }
}

// accumulate-in-place MFMA with the accumulator in the AGPR file (row tiles 0-7) or the VGPR file (row tiles 8, 9)
__device__ __forceinline__ void mfma_agpr(f32x4& c, const splitx8& a, const splitx8& b) {
    asm volatile(W4_MFMA_OP " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vgpr(f32x4& c, const splitx8& a, const splitx8& b) {
    asm volatile(W4_MFMA_OP " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
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
        const int c = (lane & 7) ^ ((row_l >> 1) & 7);
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
    const unsigned b_hi = smem_base + A_BYTES + (wn * WTN + r16) * ROWB + ((kg ^ sw) * 16);
    const unsigned b_lo = smem_base + A_BYTES + (wn * WTN + r16) * ROWB + (((4 + kg) ^ sw) * 16);
    const unsigned a_hi1 = a_hi + STAGE_BYTES, a_lo1 = a_lo + STAGE_BYTES, b_hi1 = b_hi + STAGE_BYTES, b_lo1 = b_lo + STAGE_BYTES;

    f32x4 accA[8][NT_N];      // AGPR file
    f32x4 accV[2][NT_N];      // VGPR file
    sfor<0, 8>([&](auto I) { sfor<0, NT_N>([&](auto J) { accA[decltype(I)::value][decltype(J)::value] = f32x4{0.f, 0.f, 0.f, 0.f}; }); });
    sfor<0, 2>([&](auto I) { sfor<0, NT_N>([&](auto J) { accV[decltype(I)::value][decltype(J)::value] = f32x4{0.f, 0.f, 0.f, 0.f}; }); });

    splitx8 Ah[4], Al[4];     // ring of A fragments: row tile i of a step with parity P sits in slot (2 P + i) & 3
    splitx8 Bh[NT_N], Bl[NT_N];

    const int nk = p.K / 32;

    // reads of the head of a step from the stage at `S` (compile-time 0 / 1): A tiles 0 and 1, then (tail only) the B column pairs
    auto read_a = [&](auto Sc, auto Ic, auto SLOTc) {
        constexpr int S = decltype(Sc)::value, i = decltype(Ic)::value, slot = decltype(SLOTc)::value;
        lds_rd<i * 2048>(Ah[slot], S ? a_hi1 : a_hi);
        lds_rd<i * 2048>(Al[slot], S ? a_lo1 : a_lo);
    };
    auto read_b = [&](auto Sc, auto Jc) {
        constexpr int S = decltype(Sc)::value, j = decltype(Jc)::value;
        lds_rd<j * 2048>(Bh[j], S ? b_hi1 : b_hi);
        lds_rd<j * 2048>(Bl[j], S ? b_lo1 : b_lo);
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
                if constexpr (pass == 0) mfma_agpr(accA[i][j], Al[slot], Bh[j]);
                else if constexpr (pass == 1) mfma_agpr(accA[i][j], Ah[slot], Bl[j]);
                else mfma_agpr(accA[i][j], Ah[slot], Bh[j]);
                // A fragments two row tiles ahead (ring slot of tile i - 2, retired), behind the first pass
                if constexpr (n == NT_N) read_a(IC<P>{}, IC<i + 2>{}, IC<(2 * P + i + 2) & 3>{});
                // the next stage's DMA pieces: 4,4,4,3,3 over row tiles 0..4, one piece between MFMAs
                if constexpr (i <= 4 && (n == 2 || n == 10 || n == 14 || (n == 20 && i <= 2))) {
                    constexpr int q = (i <= 2 ? 4 * i : 12 + 3 * (i - 3)) + (n == 2 ? 0 : n == 10 ? 1 : n == 14 ? 2 : 3);
                    dma(IC<q>{}, IC<q + 1>{}, (1 - P) * STAGE_BYTES, soff_next);
                }
            });
        });
        // every read of stage P has been issued; this wave's pieces of stage 1 - P have landed -> one barrier publishes and frees
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // tail: row tiles 8, 9 (VGPR accumulators) column-pair-major; the next step's A tiles 0, 1 first, then each pair's B fragments
        // as soon as its 12 MFMAs have been issued
        read_a(IC<1 - P>{}, IC<0>{}, IC<(2 * (1 - P) + 0) & 3>{});
        read_a(IC<1 - P>{}, IC<1>{}, IC<(2 * (1 - P) + 1) & 3>{});
        constexpr int s8 = (2 * P + 8) & 3, s9 = (2 * P + 9) & 3;
        sfor<0, NT_N / 2>([&](auto JPc) {
            constexpr int j0 = 2 * decltype(JPc)::value, j1 = j0 + 1;
            mfma_vgpr(accV[0][j0], Al[s8], Bh[j0]); mfma_vgpr(accV[0][j1], Al[s8], Bh[j1]);
            mfma_vgpr(accV[1][j0], Al[s9], Bh[j0]); mfma_vgpr(accV[1][j1], Al[s9], Bh[j1]);
            mfma_vgpr(accV[0][j0], Ah[s8], Bl[j0]); mfma_vgpr(accV[0][j1], Ah[s8], Bl[j1]);
            mfma_vgpr(accV[1][j0], Ah[s9], Bl[j0]); mfma_vgpr(accV[1][j1], Ah[s9], Bl[j1]);
            mfma_vgpr(accV[0][j0], Ah[s8], Bh[j0]); mfma_vgpr(accV[0][j1], Ah[s8], Bh[j1]);
            mfma_vgpr(accV[1][j0], Ah[s9], Bh[j0]); mfma_vgpr(accV[1][j1], Ah[s9], Bh[j1]);
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
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    __syncthreads();

    // ---- epilogue: per 32-row group g (row tiles 2 g, 2 g + 1) the wave transposes 32 x 128 outputs through LDS and stores row-contiguous
    // 16-byte vectors: bias / residual as float4, fp32 rows as 512-byte segments, split rows as 8-byte hi / lo groups (gemm_bf16x3.hip)
    float* scratch = reinterpret_cast<float*>(smem) + wave * (32 * EPI_PITCH);
    const int c4 = (lane & 31) * 4;
    const int col = n0 + wn * WTN + c4;
    const bool col_ok = col < p.N;
    const int colc = col_ok ? col : 0;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && col_ok) bias4 = *reinterpret_cast<const f32x4*>(p.bias + col);
    int qt = 0, qh = 0, qd = 0;
    if (p.out_mode == GEMM_OUT_QKV_HEADMAJOR) {
        const int D = p.heads * p.hd;
        qt = col / D;
        const int rem = col - qt * D;
        qh = rem / p.hd;
        qd = rem - qh * p.hd;
    }
    sfor<0, NT_M / 2>([&](auto Gc) {
        constexpr int g = decltype(Gc)::value;
        sfor<0, 2>([&](auto Tc) {
            constexpr int t = decltype(Tc)::value, ti = 2 * g + t;
            sfor<0, NT_N>([&](auto Jc) {
                constexpr int j = decltype(Jc)::value;
                f32x4 v;
                if constexpr (ti < 8) v = accA[ti][j]; else v = accV[ti - 8][j];
                // accumulator tile: register e = row 4 (lane / 16) + e, column lane % 16
#pragma unroll
                for (int e = 0; e < 4; ++e) scratch[(t * 16 + 4 * kg + e) * EPI_PITCH + j * 16 + r16] = v[e];
            });
        });
        __builtin_amdgcn_s_waitcnt(0xc07f);               // this wave's LDS writes landed (same-wave read back)
        const int row0 = m0 + wm * WTM + g * 32 + (lane >> 5);
        int qb0 = 0, qn0 = 0;
        if (p.out_mode == GEMM_OUT_QKV_HEADMAJOR) { qb0 = row0 / p.tokN; qn0 = row0 - qb0 * p.tokN; }
        constexpr int EB = 4;                              // rows per batch: LDS reads, then residual loads, then math, then stores
#pragma unroll
        for (int h0 = 0; h0 < 16; h0 += EB) {
            f32x4 v[EB];
#pragma unroll
            for (int it = 0; it < EB; ++it) v[it] = *reinterpret_cast<const f32x4*>(&scratch[((h0 + it) * 2 + (lane >> 5)) * EPI_PITCH + c4]);
            if (p.res) {
                f32x4 rs[EB];
#pragma unroll
                for (int it = 0; it < EB; ++it) rs[it] = *reinterpret_cast<const f32x4*>(p.res + (long long)min(row0 + (h0 + it) * 2, p.M - 1) * p.ldr + colc);
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
                const int row = row0 + (h0 + it) * 2;
                if (row >= p.M || !col_ok) continue;
                if (p.out_mode == GEMM_OUT_PLAIN) {
                    *reinterpret_cast<f32x4*>(p.C + (long long)row * p.ldc + col) = v[it];
                } else {
                    split_t hi[4], lo[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { hi[q] = split_hi(v[it][q]); lo[q] = split_hi(v[it][q] - (float)hi[q]); }
                    if (p.out_mode == GEMM_OUT_SPLIT_BF16) {
                        split_t* o = reinterpret_cast<split_t*>(p.Cs) + (long long)row * 2 * p.N + split_off(col, 0);
                        *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(hi);
                        *reinterpret_cast<uint2*>(o + 32) = *reinterpret_cast<const uint2*>(lo);
                    } else {   // q|k|v head-major: fp32 (exact-mode consumers) or [hi hd | lo hd] planes for the bf16x3 attention
                        int b = qb0, n = qn0 + (h0 + it) * 2;
                        while (n >= p.tokN) { n -= p.tokN; ++b; }
                        const long long rowidx = (((long long)b * 3 + qt) * p.heads + qh) * p.tokN + n;
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
        __builtin_amdgcn_s_waitcnt(0xc07f);               // read-backs done before the next group overwrites the scratch
    });
}

bool excel_gemm_w4_supported(const GemmBfArgs& p) {
    const bool vec = (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.ldr & 3) == 0 && (p.hd & 3) == 0;
    return vec && p.batch <= 1 && p.K >= 64 && (p.K % 64) == 0 && (long long)p.M * p.lda * 2 < 0x7fffffffLL && (long long)p.N * p.ldb * 2 < 0x7fffffffLL;
}

int excel_launch_gemm_w4(const GemmBfArgs& p, hipStream_t stream) {
    EXCEL_CHECK_ARG(excel_gemm_w4_supported(p), "gemm_w4: unsupported problem (vector epilogue, batch 1, K %% 64 == 0, operands below 2 GB)");
    hipLaunchKernelGGL(gemm_w4_kernel, dim3(cdiv(p.M, w4::BM) * cdiv(p.N, w4::BN)), dim3(256), 0, stream, p);
    EXCEL_CHECK_LAUNCH("gemm_w4");
    return EXCEL_OK;
}

}  // namespace EXCEL_SPLIT_NS
