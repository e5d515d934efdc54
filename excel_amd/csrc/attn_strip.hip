// Strip-resident attention-weight accumulation (bf16x3 mode) of the surgery ViT:
//     A_sum = sum_h (softmax(q.q^T s) + softmax(k.k^T s) + softmax(v.v^T s)) / 3        clip/clip_surgery_model.py:104-125,146
//     W     = sum_h softmax(q.k^T s)  (head-sum :154; head-mean for nn.MultiheadAttention, block 6)
// and the fold of W[1:,1:] into the layer-mean affinity (utils/affutils.py:180,197).
//
// One workgroup owns a STRIP of 32 query rows of one image against ALL keys.  Its waves split the keys (wave w: up to
// NTW consecutive 32-key tiles), so a complete softmax row is resident in registers: every score is computed ONCE
// (the two-pass row-statistics + recompute scheme of attn_rowpass/attn_accum computed each score twice and re-fetched
// the operand tiles per 128x64 block).  Per (head, type) "phase":
//     S^T tiles  = Y_tile . X^T         3 x v_mfma_f32_32x32x16_bf16 per 16-k step on split-bf16 q|k|v rows (bf16x3)
//     local max / exp2 / local sum      in-lane over the wave's keys (a query row = one lane) + one cross-half shuffle
//     (m_w, l_w) -> LDS, ONE barrier    global row max M = max_w m_w, row sum L = sum_w l_w 2^(m_w - M)
//     acc += p * 2^(m_w - M) / L        head reduction in registers
// Key tiles are wave-private: each wave streams its own tiles through a 2-slot LDS ring with global_load_lds (no VGPR
// round trip, counted vmcnt, no barrier); the 32-row query operand of a phase is shared and prefetched one phase ahead.
// Two sweeps over the heads reuse one accumulator set: sweep A (q.q, k.k, v.v -> A_sum), then sweep W (q.k -> W).
// Workgroups of one image are placed on one XCD (xcd_remap) so its q|k|v planes are fetched from HBM once per XCD L2.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "common.h"
#include "excel_internal.h"

namespace EXCEL_SPLIT_NS {     // compiled once per 16-bit split type (excel_internal.h, build.py)

typedef unsigned short u16;

template <class F, int... I> __device__ __forceinline__ void static_for_seq(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_seq(f, std::make_integer_sequence<int, N>{}); }

#define STRIP_VAR 3          // the variant the shipped library runs (see the kernel header)
#define GLDS(src, dst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

struct StripArgs {
    const u16* qkvs;      // split-bf16 q|k|v head-major [B,3,H,N][hi 64 | lo 64]
    u16* a_sum;           // split-bf16 [B,N][2*KP] (surgery blocks), keys [N,KP) written as zeros
    float* w_aff;         // [B,P,P] running layer-mean of W[1:,1:] (may be null)
    float* attn_out;      // [B,N,N] W of this layer (may be null)
    const float* ex_attn; // [B,P,P] LVC cue (may be null)
    int B, H, N, KP;
    int ntiles, nstrips;
    float scale;
    float w_scale;        // 1/H (head-mean) or 1 (head-sum)
    float aff_scale;      // 1/attn_layers
    float ex_scale;       // = H
    int aff_init;         // 1: w_aff = ..., 0: w_aff += ...
    int surgery;          // 1: sweep A + sweep W, 0: sweep W only
    int split_c;          // > 0: the two sweeps of a strip are separate workgroups, split_c strips per XCD (see the launcher)
    const float2* wstats; // [B,H,4,N] {row max (log2 units), 1 / row sum} of q.k from the flash row pass (type 0 slots): required by the W sweep
};

// VAR (round 4; bit set = on; STRIP_VAR is what ships):
//   bit0  fragment read-ahead: the key fragments of tile j+1 are read into the registers tile j's MFMAs have just released, one k-step
//         at a time, behind those MFMAs (needs the next tile landed: counted vmcnt(2) behind the first instalment) - the LDS round trip
//         and the wait for the tile's DMA leave the wave's stream                                                       (-0.6 % same-box)
//   bit1  static DMA cursor: which (phase, tile) the tile two ahead in the stream is, is known per unrolled tile up to one scalar
//         select on the wave's tile count - replaces the branchy run-time cursor (-77 SALU, -14 branches per phase; the kernel did
//         not notice: -0.3 %, SALU issues beside the vector stream); tile maxima by explicit v_max3_f32       (both: -1.4 % same-box)
// Measured and dropped in round 4: two accumulators per tile alternating per k-step (+7.5 %: eight more packed adds per
// tile), the softmax of tile j-1 as scalar VALU inside tile j's MFMA gaps (+30 %: unpacking adds 48 VALU per tile), and a
// producer / consumer version (waves 0-3 only DMA + MFMA, waves 4-7 only softmax / fold, score tiles through a 2-slot LDS mailbox per
// pair with polled counters: parity-green, +31 % - its skeleton alone, with MFMAs, softmax, DMA and polls removed, ran 2.5 ms against
// 1.57 ms for this kernel's; kept as tools_dev/attn_strip2_kernel.inc).  The ablation that matters: with ALL work removed (no MFMA, no
// DMA, no softmax) this kernel still takes 1.57 of its 3.43 ms - fragment reads, statistics exchange, barrier, fold and epilogue of 48
// phases; removing only the MFMAs saves 0.53, only the DMA 0.39, only the softmax 0.61.
// Two direct attempts at that skeleton lost as well (profiles/r04_ab_experiments.txt): folding the previous phase's statistics behind
// tile 0's MFMAs on EVERY wave instead of waves 4-7 (+1.3 %, 4 VGPRs spill), and a third query-strip slot so the next phase's x
// fragments are read before the phase barrier, behind the last softmax (+0.6 %).
template <int NTW, int DBG, int VAR>
__global__ __launch_bounds__(512) void attn_strip_kernel(StripArgs p) {
    constexpr int TILE_EL = 32 * 128;                                    // u16 elements of a 32-row operand tile (8 KB)
    // one LDS object: [wave][slot] key tiles (128 KB) | [parity] query strip of a phase (16 KB)
    __shared__ __attribute__((aligned(1024))) u16 ring[(8 * 2 + 2) * TILE_EL];
    u16* const xs = ring + 8 * 2 * TILE_EL;
    __shared__ float2 lstat[2 * 8 * 32];                                 // [parity][wave][q] {row max over the wave's keys (log2), sum}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int r = lane & 31, kh = lane >> 5;
    const int N = p.N, H = p.H;

    // `part`: 0 = this workgroup runs every sweep the launch asks for, 1 = sweep A only, 2 = sweep W only.  Split launch: XCD x owns
    // strips [x c, (x+1) c); its first c workgroups (dispatched first) are the long A sweeps (3H phases), the next c the short W
    // sweeps (H phases) of the same strips - longest-first, so the tail of the last round is H phases long instead of 4H.
    int id, part = 0;
    if (p.split_c > 0) {
        const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
        const int second = loc >= p.split_c ? 1 : 0;
        id = x * p.split_c + loc - second * p.split_c;
        part = 1 + second;
        if (id >= p.B * p.nstrips) return;                            // padding of the last XCD's chunk (whole workgroup, before any barrier)
    } else {
        id = xcd_remap(blockIdx.x, gridDim.x);
    }
    const int b = id / p.nstrips, strip = id - b * p.nstrips;
    const int q0 = strip * 32;

    // this wave's key tiles [first, first + cnt); cnt is NTW or NTW - 1 (host guarantees it)
    const int tb_ = p.ntiles / nw, tr_ = p.ntiles - tb_ * nw;
    const int cnt = tb_ + (wave < tr_ ? 1 : 0);
    const int first = wave * tb_ + min(wave, tr_);
    const bool full = cnt == NTW;

    u16* myring = ring + wave * 2 * TILE_EL;
    const unsigned ring_addr = lds_addr(myring), xs_addr = lds_addr(xs), lstat_addr = lds_addr(lstat);

    // LDS-DMA through a buffer descriptor over this image's q|k|v planes: per-lane byte offsets are loop invariant (4 VGPRs), the
    // tile / plane position goes into the scalar offset.  One 1-KB wave instruction = 4 rows x 16 chunks of 16 B; chunk c of row rr
    // lands at slot c ^ (rr & 15) (swizzle on the SOURCE address, the LDS image is lane-linear).  Rows past the end of a plane read
    // the next plane (finite data, masked below) or, at the end of the image, the descriptor's out-of-range zeros.
    const long long img_el = 3LL * H * N * 128;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.qkvs + (long long)b * img_el), 0, (int)(img_el * 2), 0x00020000);
    int voff[4];
    {
        const int srow = lane >> 4, sbase = (lane & 15) ^ srow;      // (lane&15) ^ ((4i + srow) & 15) = sbase ^ 4(i & 3)
#pragma unroll
        for (int i = 0; i < 4; ++i) voff[i] = (4 * i + srow) * 256 + ((sbase ^ (4 * i)) * 16);
    }
    auto plane_off = [&](int t, bool xside) -> int {                // byte offset of the (type, head) plane a phase reads
        int typ, h;
        if (t < 3 * H) { h = t / 3; typ = t - 3 * h; }              // q.q, k.k, v.v
        else { h = t - 3 * H; typ = xside ? 0 : 1; }                // q (rows) . k (keys)
        return (typ * H + h) * N * 256;
    };
    typedef __attribute__((address_space(3))) unsigned char* lds_bptr;
    // pieces [i0, i1) of an 8-piece (8 KB) tile
    auto dma_pieces = [&](int soff, unsigned dst_byte, int i0, int i1) {
        // opaque to the optimiser: otherwise the 16 per-instruction LDS destinations (m0 values) are hoisted out of the loop
        // into SGPRs that spill (v_readlane + hazard nops inside the loop); recomputed here they are one s_add each
        asm volatile("" : "+s"(dst_byte));
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i >= i0 && i < i1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_bptr)(unsigned long long)(dst_byte + i * 1024), 16, voff[i & 3], soff + (i >> 2) * 4096, 0, 0);
    };
    auto dma_tile = [&](int soff, unsigned dst_byte) { dma_pieces(soff, dst_byte, 0, 8); };

    // ---- issue cursor over (phase, tile) in consumption order.  Past the end of a sweep it keeps re-loading the last tile into
    // the slot just freed (never read again), so every wait in the loop is the same counted vmcnt(8): no end-of-stream branches.
    int it_t, it_j, it_g, it_t1, it_poff;
    auto issue_advance = [&]() {
        ++it_g;
        if (it_t < it_t1 - 1 || it_j < cnt - 1) {
            if (++it_j == cnt) { it_j = 0; ++it_t; it_poff = plane_off(it_t, false); }
        }
    };
    auto issue_next = [&]() {
        dma_tile(it_poff + (first + it_j) * 8192, ring_addr + (it_g & 1) * (TILE_EL * 2));
        issue_advance();
    };
    // the same tile in four instalments of two pieces, placed between the MFMA groups of the tile being multiplied: eight pieces
    // issued back to back by every wave at once queue up in the CU's one address path and hold the wave off its MFMAs
    auto issue_next_part = [&](int part) {
        dma_pieces(it_poff + (first + it_j) * 8192, ring_addr + (it_g & 1) * (TILE_EL * 2), 2 * part, 2 * part + 2);
        if (part == 3) issue_advance();
    };
    auto issue_x = [&](int t, int par) {
        const int soff = plane_off(t, true) + q0 * 256;
        unsigned xb = xs_addr + par * (TILE_EL * 2);
        asm volatile("" : "+s"(xb));
        for (int pi = wave; pi < 8; pi += nw)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_bptr)(unsigned long long)(xb + pi * 1024), 16, voff[pi & 3], soff + (pi >> 2) * 4096, 0, 0);
    };

    // Known row statistics (W sweep, round 5): the flash row pass of the same layer has already reduced every q.k row (it needs max and
    // sum for the attention output) and left {M, 1/L} in `wstats`.  With them a W phase needs no running maximum, no rescale, no
    // per-tile reference, no exchange of (m, l) between the waves and no deferred fold: p = 2^(s c2 - M) goes into the accumulators as
    // acc += p / L right behind the tile's MFMAs.  The 32 pairs of a phase (256 B) ride with the query strip: one 4-byte-per-lane LDS-DMA
    // instruction of the wave that issues the fewest strip pieces, into the (otherwise unused) exchange slot of the phase's parity.
    const __amdgpu_buffer_rsrc_t rsrc_st = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wstats ? p.wstats + (long long)b * H * 4 * N : nullptr), 0,
                                                                             p.wstats ? H * 4 * N * 8 : 0, 0x00020000);
    auto issue_st = [&](int t, int par) {
        if (wave != nw - 1) return;
        const int h = t - 3 * H;
        unsigned sb = lstat_addr + par * (8 * 32 * 8);
        asm volatile("" : "+s"(sb));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_st, (lds_bptr)(unsigned long long)sb, 4, lane * 4, ((h * 4) * N + q0) * 8, 0, 0);
    };

    // VAR bit 1: static cursor.  While tile j of phase t is multiplied, the tile two ahead in the stream goes out: (t, j + 2) if the wave
    // has that many tiles, else (t + 1, j + 2 - cnt); poff_cur / poff_nxt are the plane offsets of phases t and min(t + 1, t1 - 1).
    int poff_cur = 0, poff_nxt = 0;
    auto issue_part_static = [&](int j, int part, int gcnt) {
        const bool same = j + 2 < cnt;
        const int tix = same ? j + 2 : j + 2 - cnt;
        dma_pieces((same ? poff_cur : poff_nxt) + (first + tix) * 8192, ring_addr + (gcnt & 1) * (TILE_EL * 2), 2 * part, 2 * part + 2);
    };

    const float c2 = p.scale * 1.4426950408889634f;              // scores enter the softmax in log2 units
    const int last_tile = p.ntiles - 1;
    const bool ragged = (N & 31) != 0;
    const int nvalid_last = N - last_tile * 32 - 4 * kh;         // accumulator row (e&3)+8(e>>2) of the last tile is a real key iff < this

    f32x16 acc[NTW];

    // One sweep over phases [t0, t1).  Program order per phase (the softmax block written after a tile's MFMAs works on the PREVIOUS
    // tile, so it issues in their shadow: an MFMA occupies the matrix pipe for 32 cycles, a few VALU issue slots):
    //     X fragments | tile j: MFMA || softmax(tile j-1) | softmax(last tile) | stats -> barrier | factors, accumulate
    // Softmax is "online" per LANE (a lane owns 16 keys of every tile of its query row): tile j is exponentiated against the running
    // maximum m_j of tiles 0..j (kept per tile), the lane's sum is rescaled when the maximum moves; the two lane halves of a row
    // merge, every wave publishes (m, l) and the factor of tile j becomes 2^(m_j - M) / L with the global M, L: exact, single pass.
    // Measured (profiles/): the waves of a strip run in lock step and are bound by their own in-order instruction streams, so the
    // VALU work is written with packed fp32 operations (v_pk_fma_f32, v_pk_add_f32) and the statistics exchange is one LDS batch.
    auto run_sweep = [&](int t0, int t1, auto known_c) {
        constexpr bool KNOWN = decltype(known_c)::value;           // row statistics come from `wstats` (W sweep behind the flash row pass)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        it_t = t0; it_j = 0; it_g = 0; it_t1 = t1; it_poff = plane_off(t0, false);
        issue_x(t0, 0);
        if constexpr (KNOWN) issue_st(t0, 0);
        issue_x(min(t0 + 1, t1 - 1), 1);
        if constexpr (KNOWN) issue_st(min(t0 + 1, t1 - 1), 1);
        issue_next();
        issue_next();
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");         // both query strips landed (the two key tiles may be in flight)
        __builtin_amdgcn_s_barrier();                              // ... and are visible to every wave
        int gc = 0;

        // The probabilities of a phase are folded into the accumulators (block F: all waves' (m, l) -> M, L -> per-tile factors ->
        // acc += p f) INSIDE the next phase, and the two waves that share a SIMD (w, w + 4) do it on opposite sides of tile 0's
        // MFMAs: F is ~200 cycles of VALU work, so the pair leaves the barrier half a tile out of step and stays that way - while
        // one wave's 12 dependent MFMAs own the matrix pipe the other runs its softmax on the VALU.  In step (round 2), both waves
        // of a SIMD queued their MFMA blocks, then their softmax blocks, on one pipe each (matrix pipe 25 % busy, VALU 38 %).
        f32x16 s[NTW];
        float mref[NTW];                                           // running maximum (log2 units) tile j was exponentiated against
        bool pending = false;
        const bool late_f = (wave & 4) != 0;
        splitx8 yh[4], yl[4];                                       // key fragments (VAR bit 0: carried from tile to tile)
        if constexpr ((VAR & 1) != 0) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // tile 0 of the stream landed (tile 1 may be in flight)
            const unsigned k0 = ring_addr + (r * 128) * 2;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                yh[s4] = lds_read16(k0 + (((s4 * 2 + kh) ^ (r & 15)) * 16));
                yl[s4] = lds_read16(k0 + (((8 + s4 * 2 + kh) ^ (r & 15)) * 16));
            }
        }
        auto apply_phase = [&](int tp) {
            const unsigned ls = lstat_addr + (tp & 1) * (8 * 32 * 8);
            float M, L = 0.f;
            {
                float2 st[8];
#pragma unroll
                for (int w2 = 0; w2 < 8; ++w2) st[w2] = lds_read8(ls + r * 8 + (w2 < nw ? w2 : 0) * (32 * 8));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]), "+v"(st[4]), "+v"(st[5]), "+v"(st[6]), "+v"(st[7])::"memory");
                M = fmaxf(fmaxf(st[0].x, st[1].x), st[2].x);                              // slots >= nw repeat slot 0
                M = fmaxf(fmaxf(M, st[3].x), st[4].x);
                M = fmaxf(fmaxf(M, st[5].x), fmaxf(st[6].x, st[7].x));
#pragma unroll
                for (int w2 = 0; w2 < 8; ++w2)
                    if (w2 < nw) L = fmaf(st[w2].y, __builtin_amdgcn_exp2f(st[w2].x - M), L);
            }
            const float rl = 1.f / L;
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                if (j < NTW - 1 || full) {
                    const float f = __builtin_amdgcn_exp2f(mref[j] - M) * rl;
                    const f32x2 f2 = {f, f};
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        const f32x2 a = __builtin_elementwise_fma(f32x2{s[j][e], s[j][e + 1]}, f2, f32x2{acc[j][e], acc[j][e + 1]});
                        acc[j][e] = a[0];
                        acc[j][e + 1] = a[1];
                    }
                }
        };
        for (int t = t0; t < t1; ++t) {
            const int par = (t - t0) & 1;
            if constexpr ((VAR & 2) != 0) {
                poff_cur = (t == t0) ? plane_off(t0, false) : poff_nxt;
                poff_nxt = plane_off(min(t + 1, t1 - 1), false);
            }
            // query-strip fragments (B operand): row r, k-step s4 -> chunk (2 s4 + kh) of hi, 8 + (2 s4 + kh) of lo
            splitx8 xh[4], xl[4];
            {
                const unsigned xr = xs_addr + (par * TILE_EL + r * 128) * 2;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    xh[s4] = lds_read16(xr + (((s4 * 2 + kh) ^ (r & 15)) * 16));
                    xl[s4] = lds_read16(xr + (((8 + s4 * 2 + kh) ^ (r & 15)) * 16));
                }
                lds_wait8(xh, xl);
            }
            // known statistics of this phase's rows: {M, 1/L} of query row r
            f32x2 nm2 = {0.f, 0.f}, li2 = {0.f, 0.f};
            if constexpr (KNOWN) {
                float2 st = lds_read8(lstat_addr + par * (8 * 32 * 8) + r * 8);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(st)::"memory");
                nm2 = f32x2{-st.x, -st.x};
                li2 = f32x2{st.y, st.y};
            }
            if (pending && !late_f) apply_phase(t - 1);            // early F: before tile 0
            // finite "minus infinity": a lane whose 16 keys of the (ragged) last tile are all padding must not form (-inf) - (-inf)
            float m_run = -1e30f, l_run = 0.f;
            auto softmax_tile = [&](int j) __attribute__((always_inline)) {
                if constexpr (KNOWN) {                             // acc += 2^(s c2 - M) / L, nothing carried
                    const f32x2 c22 = {c2, c2};
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        const f32x2 a = __builtin_elementwise_fma(f32x2{s[j][e], s[j][e + 1]}, c22, nm2);
                        const f32x2 pe = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                        const f32x2 o = __builtin_elementwise_fma(pe, li2, f32x2{acc[j][e], acc[j][e + 1]});
                        acc[j][e] = o[0];
                        acc[j][e + 1] = o[1];
                    }
                    return;
                }
                float tm;
                if constexpr ((VAR & 2) != 0) {
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tm) : "v"(s[j][0]), "v"(s[j][1]), "v"(s[j][2]));
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(tm) : "v"(s[j][3]), "v"(s[j][4]));
#pragma unroll
                    for (int e = 5; e < 15; e += 2) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(tm) : "v"(s[j][e]), "v"(s[j][e + 1]));
                    asm("v_max_f32 %0, %0, %1" : "+v"(tm) : "v"(s[j][15]));
                } else {
                    tm = fmaxf(s[j][0], s[j][1]);
#pragma unroll
                    for (int e = 2; e < 16; e += 2) tm = fmaxf(fmaxf(tm, s[j][e]), s[j][e + 1]);   // v_max3_f32
                }
                const float m_new = fmaxf(m_run, tm * c2);
                l_run *= __builtin_amdgcn_exp2f(m_run - m_new);    // first tile: 2^(-inf) = 0
                const f32x2 c22 = {c2, c2}, nm2 = {-m_new, -m_new};
                f32x2 ps2 = {0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 a = __builtin_elementwise_fma(f32x2{s[j][e], s[j][e + 1]}, c22, nm2);
                    const f32x2 pe = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                    s[j][e] = pe[0];
                    s[j][e + 1] = pe[1];
                    ps2 += pe;
                }
                l_run += ps2[0] + ps2[1];
                m_run = m_new;
                mref[j] = m_new;
            };
            {
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    // (one call site per tile index for the softmax: `full ? tile NTW-1 : tile NTW-2` behind the loop let the optimiser
                    //  fold the two sites into one body with a run-time tile index - and the accumulators went to scratch)
                    const bool have = j < NTW - 1 || full;
                    f32x16 sj;
                    if (have) {
                        if constexpr ((VAR & 1) == 0) {
                            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                    // tile gc landed (gc+1 may be in flight)
                            const unsigned kr = ring_addr + ((gc & 1) * TILE_EL + r * 128) * 2;
    #pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                yh[s4] = lds_read16(kr + (((s4 * 2 + kh) ^ (r & 15)) * 16));
                                yl[s4] = lds_read16(kr + (((8 + s4 * 2 + kh) ^ (r & 15)) * 16));
                            }
                        }
                        lds_wait8(yh, yl);                                                      // fragments in registers: the slot is free
                        const unsigned kn = ring_addr + (((gc + 1) & 1) * TILE_EL + r * 128) * 2;  // (VAR bit 0) the next tile of the stream
                        if ((DBG & 1) && !(DBG & 2)) issue_next();                              // tile gc+2 -> this slot
                        ++gc;
    #pragma unroll
                        for (int e = 0; e < 16; ++e) sj[e] = 0.f;
                        if (!(DBG & 1)) {
    #pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                if constexpr ((VAR & 2) != 0) issue_part_static(j, s4, gc + 1);     // (gc was advanced: this tile's slot)
                                else if (!(DBG & 2)) issue_next_part(s4);                       // tile gc+2 -> this slot, two pieces per k-step
                                sj = EXCEL_MFMA16(yl[s4], xh[s4], sj, 0, 0, 0);
                                sj = EXCEL_MFMA16(yh[s4], xl[s4], sj, 0, 0, 0);
                                sj = EXCEL_MFMA16(yh[s4], xh[s4], sj, 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);                              // keep the instalments where they are
                                if constexpr ((VAR & 1) != 0) {
                                    // bit 0: the next tile's fragments of this k-step into the registers these MFMAs have just read
                                    if (s4 == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // all but the two pieces just issued: it landed
                                    yh[s4] = lds_read16(kn + (((s4 * 2 + kh) ^ (r & 15)) * 16));
                                    yl[s4] = lds_read16(kn + (((8 + s4 * 2 + kh) ^ (r & 15)) * 16));
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        } else {
                            sj[0] = (float)yl[0][0] + (float)yh[3][1] + (float)xh[0][0] + (float)xl[3][1];
                        }
                        if constexpr (!KNOWN) {
                            if (NTW < 5) {                                                      // (at 5 tiles per wave the register file has no room for the overlap)
                                if (j > 0 && !(DBG & 4)) softmax_tile(j - 1);                   // in the shadow of these MFMAs
                            }
                        }
                    }
                    if constexpr (KNOWN) {                                                      // (a wave without tile NTW-1 folds its last tile here)
                        if (NTW < 5 && j > 0 && !(DBG & 4)) softmax_tile(j - 1);
                    }
                    if (have) {
                        if (ragged && first + j == last_tile) {                                 // wave-uniform: keys >= N only here
                            int lim = nvalid_last;                    // opaque: the 16 lane masks must not be hoisted into (spilled) SGPR pairs
                            asm volatile("" : "+v"(lim));
    #pragma unroll
                            for (int e = 0; e < 16; ++e)
                                if ((e & 3) + 8 * (e >> 2) >= lim) sj[e] = -INFINITY;
                        }
                        if (j == 0 && pending && late_f) apply_phase(t - 1);                    // late F: behind tile 0's MFMAs, before s[0] is replaced
                        s[j] = sj;
                    }
                }
}
            if (!(DBG & 4)) {
                if (NTW < 5) {
                    if (full) softmax_tile(NTW - 1);
                    else if constexpr (!KNOWN) { if (NTW > 1) softmax_tile(NTW > 1 ? NTW - 2 : 0); }     // (KNOWN: folded inside the loop)
                } else {
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        if (j < NTW - 1 || full) softmax_tile(j);
                }
            } else {
                m_run = 0.f; l_run = 1.f;
#pragma unroll
                for (int j = 0; j < NTW; ++j) mref[j] = 0.f;
            }
            // the two lane halves of a query row merge their (m, l); one entry per wave goes to the exchange
            if constexpr (!KNOWN) {
                float m_a, m_b, l_a, l_b;                     // (a: lanes 0..31, b: lanes 32..63 - in every lane, no LDS round trip)
                wave_halves(m_run, m_a, m_b);
                wave_halves(l_run, l_a, l_b);
                const float m_w = fmaxf(m_a, m_b);
                l_run = l_a * __builtin_amdgcn_exp2f(m_a - m_w) + l_b * __builtin_amdgcn_exp2f(m_b - m_w);
                m_run = m_w;
            }
            const unsigned ls = lstat_addr + (t & 1) * (8 * 32 * 8);
            if (!KNOWN && kh == 0) lds_write8(ls + (wave * 32 + r) * 8, make_float2(m_run, l_run));
            // the query strip of phase t+1 (issued one phase ago) must have landed before the barrier publishes it
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                    // (a key tile was issued after it)
            __builtin_amdgcn_s_barrier();                                                   // raw: no vmcnt(0) drain of the tile stream
            if (!(DBG & 2)) issue_x(min(t + 2, t1 - 1), par);                               // slot of phase t: every wave has its fragments
            if constexpr (KNOWN) { if (!(DBG & 2)) issue_st(min(t + 2, t1 - 1), par); }     // (its statistics have been read as well)
            else if (NTW < 5) pending = true;                                               // F(t) runs inside phase t+1 (or after the loop)
            else apply_phase(t);                                                            // (5 tiles per wave: no registers to carry a phase)
        }
        if (pending) apply_phase(t1 - 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };

    // ---- epilogues: per 32x32 tile, transpose [key][q] -> [q][key] through this wave's (drained) ring memory
    float* tbuf = reinterpret_cast<float*>(myring);                 // 32 x 36 floats
    constexpr int TP = 36;
    auto to_lds = [&](const f32x16& a, float mul) {
#pragma unroll
        for (int e = 0; e < 16; ++e) tbuf[r * TP + c32_row(e, lane)] = a[e] * mul;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    if (p.surgery && part != 2) {
        run_sweep(0, 3 * H, std::false_type{});
        // A_sum: split-bf16 rows; one 32-key tile = one 128-B block [hi 32 | lo 32]
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (j >= cnt) continue;
            const int tile = first + j;
            to_lds(acc[j], 1.f / 3.f);
#pragma unroll
            for (int itr = 0; itr < 2; ++itr) {
                const int ch = itr * 64 + lane;                 // 128 chunks of 8 keys
                const int qq = ch >> 2, g8 = ch & 3;
                const int qg = q0 + qq;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(tbuf + qq * TP + g8 * 8);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(tbuf + qq * TP + g8 * 8 + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (p.ex_attn && qg >= 1 && qg < N) {
                    // LVC branch (clip_surgery_model.py:140-141): every head's attn[1:,1:] += ex_attn -> head-sum gains H x ex_attn
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int kg = tile * 32 + g8 * 8 + k;
                        if (kg >= 1 && kg < N) v[k] += p.ex_scale * p.ex_attn[((long long)b * (N - 1) + (qg - 1)) * (N - 1) + (kg - 1)];
                    }
                }
                if (qg < N) {
                    splitx8 hi, lo;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        hi[k] = split_hi(v[k]);
                        lo[k] = split_hi(v[k] - (float)hi[k]);
                    }
                    u16* o = p.a_sum + ((long long)b * N + qg) * 2 * p.KP + tile * 64 + g8 * 8;
                    *reinterpret_cast<splitx8*>(o) = hi;
                    *reinterpret_cast<splitx8*>(o + 32) = lo;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }

    if ((p.w_aff || p.attn_out) && part != 1) {
        run_sweep(3 * H, 4 * H, std::true_type{});              // (the launcher insists on wstats: a third copy of the sweep for the exchange form spills)
        const long long P = N - 1;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (j >= cnt) continue;
            const int kg = (first + j) * 32 + r;
            to_lds(acc[j], p.w_scale);
            float oldw[16];
            const bool rmw = p.w_aff && !p.aff_init;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qg = q0 + 2 * i + kh;
                oldw[i] = (rmw && qg < N && qg >= 1 && kg >= 1 && kg < N) ? p.w_aff[((long long)b * P + (qg - 1)) * P + (kg - 1)] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qq = 2 * i + kh, qg = q0 + qq;
                const float pw = tbuf[qq * TP + r];
                if (qg >= N || kg >= N) continue;
                if (p.attn_out) p.attn_out[((long long)b * N + qg) * N + kg] = pw;
                if (p.w_aff && qg >= 1 && kg >= 1) p.w_aff[((long long)b * P + (qg - 1)) * P + (kg - 1)] = oldw[i] + pw * p.aff_scale;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

// Strip-resident accumulate pass; returns EXCEL_ERR_ARG-free "not applicable" (1) when the shape does not fit (caller falls back).
bool excel_attn_strip_supported(int N) { return cdiv(N, 32) <= 40; }       // 8 waves x 5 key tiles: beyond that the two-pass kernels run

int excel_launch_attn_strip(const unsigned short* qkvs, unsigned short* a_sum, float* w_aff, float* attn_out, int B, int H, int N,
                            int KP, int hd, float scale, int surgery, float w_scale, float aff_scale, int aff_init, const float* ex_attn,
                            hipStream_t st, const float* wstats) {
    EXCEL_CHECK_ARG(hd == 64, "attention: head_dim must be 64 (got %d)", hd);
    EXCEL_CHECK_ARG(qkvs && (!surgery || (a_sum && KP == cdiv(N, 32) * 32)), "attn_strip: bad a_sum/KP");
    const int ntiles = cdiv(N, 32);
    EXCEL_CHECK_ARG(wstats || !(w_aff || attn_out), "attn_strip: the W sweep needs the flash row pass's q.k row statistics (wstats)");
    EXCEL_CHECK_ARG(excel_attn_strip_supported(N), "attn_strip: N=%d exceeds the strip-resident envelope (ask excel_attn_strip_supported)", N);
    ProfScope prof__(PROF_ATTN_ACCUM, st);
    const int ntw = cdiv(ntiles, 8);
    const int nw = cdiv(ntiles, ntw);                          // 25 tiles: 7 waves x (4,4,4,4,3,3,3)
    EXCEL_CHECK_ARG(ntiles / nw >= ntw - 1 && (long long)3 * H * N * 256 < (1LL << 31), "attn_strip: unsupported shape");
    StripArgs a{qkvs, a_sum, w_aff, attn_out, surgery ? ex_attn : nullptr, B, H, N, KP, ntiles, cdiv(N, 32), scale, w_scale, aff_scale, (float)H,
                aff_init, surgery, 0, reinterpret_cast<const float2*>(wstats)};
    // Both sweeps wanted: one workgroup per (strip, sweep).  A strip workgroup fills a CU (148 KB LDS), so B x nstrips = 800 uniform
    // workgroups on 256 CUs are 3.125 rounds = 4 rounds of 4H phases; split, the 3H-phase workgroups go first and the H-phase ones
    // level the tail: 150-156 phase-times per CU instead of 192.
    const bool split = surgery && (w_aff || attn_out);
    if (split) a.split_c = cdiv(B * a.nstrips, 8);
    const dim3 grid(split ? 8 * 2 * a.split_c : B * a.nstrips), block(nw * 64);
#ifdef EXCEL_DEV
    // dev build only: ablation variants (bit0 no MFMA, bit1 no DMA in the loop, bit2 no softmax, bit3 no schedule groups)
    static const int dbg = getenv("EXCEL_STRIP_DBG") ? atoi(getenv("EXCEL_STRIP_DBG")) : 0;
    if (dbg && ntw == 4) {
        switch (dbg) {
            case 1: hipLaunchKernelGGL((attn_strip_kernel<4, 1, 0>), grid, block, 0, st, a); break;
            case 2: hipLaunchKernelGGL((attn_strip_kernel<4, 2, 0>), grid, block, 0, st, a); break;
            case 3: hipLaunchKernelGGL((attn_strip_kernel<4, 3, 0>), grid, block, 0, st, a); break;
            case 4: hipLaunchKernelGGL((attn_strip_kernel<4, 4, 0>), grid, block, 0, st, a); break;
            case 5: hipLaunchKernelGGL((attn_strip_kernel<4, 5, 0>), grid, block, 0, st, a); break;
            case 6: hipLaunchKernelGGL((attn_strip_kernel<4, 6, 0>), grid, block, 0, st, a); break;
            case 7: hipLaunchKernelGGL((attn_strip_kernel<4, 7, 0>), grid, block, 0, st, a); break;
            default: hipLaunchKernelGGL((attn_strip_kernel<4, 8, 0>), grid, block, 0, st, a); break;
        }
        EXCEL_CHECK_LAUNCH("attn_strip");
        return EXCEL_OK;
    }
#endif
#ifdef EXCEL_DEV
    static const int var = getenv("EXCEL_STRIP_VAR") ? atoi(getenv("EXCEL_STRIP_VAR")) : STRIP_VAR;
    if (ntw == 4 && var != STRIP_VAR) {
        switch (var) {
            case 0: hipLaunchKernelGGL((attn_strip_kernel<4, 0, 0>), grid, block, 0, st, a); break;
            case 1: hipLaunchKernelGGL((attn_strip_kernel<4, 0, 1>), grid, block, 0, st, a); break;
            default: hipLaunchKernelGGL((attn_strip_kernel<4, 0, 2>), grid, block, 0, st, a); break;
        }
        EXCEL_CHECK_LAUNCH("attn_strip");
        return EXCEL_OK;
    }
#endif
    switch (ntw) {
        // (the round-4 variants for 3 and 4 tiles per wave: the static cursor needs >= 2 tiles per wave, 5 tiles have no register room)
        case 1: hipLaunchKernelGGL((attn_strip_kernel<1, 0, 0>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((attn_strip_kernel<2, 0, 0>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((attn_strip_kernel<3, 0, STRIP_VAR>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((attn_strip_kernel<4, 0, STRIP_VAR>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((attn_strip_kernel<5, 0, 0>), grid, block, 0, st, a); break;
    }
    EXCEL_CHECK_LAUNCH("attn_strip");
    return EXCEL_OK;
}

}  // namespace EXCEL_SPLIT_NS
