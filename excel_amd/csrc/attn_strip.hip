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
#include "common.h"
#include "excel_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#define GLDS(src, dst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// LDS accesses of the streaming loop are written as inline asm: the compiler's wait-count pass treats every ds_read as a
// possible reader of a pending global_load_lds and drains the whole DMA queue (s_waitcnt vmcnt(0)) in front of it, which
// serialises the tile stream (that is what held attn_accum_bf_kernel at ~20 % matrix-core busy).  The waits here are explicit.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ bf16x8 lds_read16(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ float2 lds_read8(unsigned addr) {
    float2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write8(unsigned addr, float2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// s_waitcnt lgkmcnt(0) that the consumers of the eight fragments depend on (keeps the MFMAs behind the wait)
__device__ __forceinline__ void lds_wait8(bf16x8 (&a)[4], bf16x8 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])::"memory");
}

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
};

template <int NTW>
__global__ __launch_bounds__(512) void attn_strip_kernel(StripArgs p) {
    constexpr int TILE_EL = 32 * 128;                                    // u16 elements of a 32-row operand tile (8 KB)
    // one LDS object: [wave][slot] key tiles (128 KB) | [parity] query strip of a phase (16 KB)
    __shared__ __attribute__((aligned(1024))) u16 ring[(8 * 2 + 2) * TILE_EL];
    u16* const xs = ring + 8 * 2 * TILE_EL;
    __shared__ float2 lstat[2 * 8 * 32];                                 // [parity][wave][q] {local max (log2), local sum}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int r = lane & 31, kh = lane >> 5;
    const int N = p.N, H = p.H;

    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int b = id / p.nstrips, strip = id - b * p.nstrips;
    const int q0 = strip * 32;

    // this wave's key tiles [first, first + cnt)
    const int tb_ = p.ntiles / nw, tr_ = p.ntiles - tb_ * nw;
    const int cnt = tb_ + (wave < tr_ ? 1 : 0);
    const int first = wave * tb_ + min(wave, tr_);

    u16* myring = ring + wave * 2 * TILE_EL;
    const unsigned ring_addr = lds_addr(myring), xs_addr = lds_addr(xs), lstat_addr = lds_addr(lstat);
    // staging map of one 1-KB wave instruction: 4 rows x 16 chunks of 16 B; chunk c of row rr sits at slot c ^ (rr & 15)
    const int srow = lane >> 4;                                  // row within the 4-row piece
    const int sbase = (lane & 15) ^ srow;                        // (lane&15) ^ ((4i + srow) & 15) = sbase ^ (4 (i & 3))

    auto phase_desc = [&](int t, int& h, int& tx, int& ty) {
        if (t < 3 * H) { h = t / 3; const int ty3 = t - 3 * h; tx = ty3; ty = ty3; }   // q.q, k.k, v.v
        else { h = t - 3 * H; tx = 0; ty = 1; }                                          // q.k
    };
    auto plane = [&](int typ, int h) { return p.qkvs + (((long long)b * 3 + typ) * H + h) * (long long)N * 128; };

    // ---- issue cursor over (phase, tile) in consumption order
    int it_t, it_j, it_g, it_t1;
    auto issue_next = [&]() -> bool {
        if (it_t >= it_t1) return false;
        int h, tx, ty;
        phase_desc(it_t, h, tx, ty);
        const u16* Y = plane(ty, h);
        const int key0 = (first + it_j) * 32;
        u16* dst = myring + (it_g & 1) * TILE_EL;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = min(key0 + 4 * i + srow, N - 1);
            GLDS(Y + (long long)key * 128 + ((sbase ^ (4 * (i & 3))) * 8), dst + i * 512);
        }
        ++it_g;
        if (++it_j == cnt) { it_j = 0; ++it_t; }
        return true;
    };
    auto issue_x = [&](int t, int par) {
        int h, tx, ty;
        phase_desc(t, h, tx, ty);
        const u16* X = plane(tx, h);
        for (int pi = wave; pi < 8; pi += nw) {
            const int q = min(q0 + 4 * pi + srow, N - 1);
            GLDS(X + (long long)q * 128 + ((sbase ^ (4 * (pi & 3))) * 8), xs + par * TILE_EL + pi * 512);
        }
    };

    const float c2 = p.scale * 1.4426950408889634f;              // scores enter the softmax in log2 units
    const int last_tile = p.ntiles - 1;
    const bool ragged = (N & 31) != 0;

    f32x16 acc[NTW];

    auto run_sweep = [&](int t0, int t1) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        it_t = t0; it_j = 0; it_g = 0; it_t1 = t1;
        const int total = (t1 - t0) * cnt;
        issue_x(t0, 0);
        if (t0 + 1 < t1) issue_x(t0 + 1, 1);
        int ntile_issued = 0;
        if (issue_next()) ++ntile_issued;
        if (issue_next()) ++ntile_issued;
        if (ntile_issued == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (ntile_issued == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // both query strips visible to every wave

        int gc = 0;
        for (int t = t0; t < t1; ++t) {
            const int par = (t - t0) & 1;
            // query-strip fragments (B operand): row r, k-step s4 -> chunk (2 s4 + kh) of hi, 8 + (2 s4 + kh) of lo
            bf16x8 xh[4], xl[4];
            {
                const unsigned xr = xs_addr + (par * TILE_EL + r * 128) * 2;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    xh[s4] = lds_read16(xr + (((s4 * 2 + kh) ^ (r & 15)) * 16));
                    xl[s4] = lds_read16(xr + (((8 + s4 * 2 + kh) ^ (r & 15)) * 16));
                }
                lds_wait8(xh, xl);
            }
            f32x16 s[NTW];
            bool issued_here = false;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                if (j < cnt) {
                    if (gc + 1 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile gc landed (gc+1 may be in flight)
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const unsigned kr = ring_addr + ((gc & 1) * TILE_EL + r * 128) * 2;
                    bf16x8 yh[4], yl[4];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        yh[s4] = lds_read16(kr + (((s4 * 2 + kh) ^ (r & 15)) * 16));
                        yl[s4] = lds_read16(kr + (((8 + s4 * 2 + kh) ^ (r & 15)) * 16));
                    }
                    lds_wait8(yh, yl);                                                      // fragments in registers: the slot is free
                    issued_here |= issue_next();                                            // tile gc+2 -> this slot
                    ++gc;
                    f32x16 sj;
#pragma unroll
                    for (int e = 0; e < 16; ++e) sj[e] = 0.f;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        sj = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yl[s4], xh[s4], sj, 0, 0, 0);
                        sj = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[s4], xl[s4], sj, 0, 0, 0);
                        sj = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[s4], xh[s4], sj, 0, 0, 0);
                    }
                    if (ragged && first + j == last_tile) {                                 // wave-uniform: keys >= N only here
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            if (last_tile * 32 + c32_row(e, lane) >= N) sj[e] = -INFINITY;
                    }
                    s[j] = sj;
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[j][e] = -INFINITY;
                }
            }
            // local softmax statistics over this wave's keys (query r lives in lanes r and r+32)
            float mx = s[0][0];
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[j][e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s[j][e] = __builtin_amdgcn_exp2f(fmaf(s[j][e], c2, -mx));
                    ps += s[j][e];
                }
            ps += __shfl_xor(ps, 32, 64);
            const unsigned ls = lstat_addr + (t & 1) * 8 * 32 * 8;
            if (kh == 0) lds_write8(ls + (wave * 32 + r) * 8, make_float2(mx, ps));
            // the query strip of phase t+1 (issued one phase ago) must have landed before the barrier publishes it
            if (issued_here) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                                   // raw: no vmcnt(0) drain of the tile stream
            if (t + 2 < t1) issue_x(t + 2, par);                                            // slot of phase t: every wave has its fragments
            float2 st8[8];
#pragma unroll
            for (int w2 = 0; w2 < 8; ++w2) st8[w2] = lds_read8(ls + ((w2 < nw ? w2 : 0) * 32 + r) * 8);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(st8[0]), "+v"(st8[1]), "+v"(st8[2]), "+v"(st8[3]), "+v"(st8[4]), "+v"(st8[5]), "+v"(st8[6]), "+v"(st8[7])::"memory");
            float M = st8[0].x;
#pragma unroll
            for (int w2 = 1; w2 < 8; ++w2) M = fmaxf(M, st8[w2].x);          // slots >= nw repeat wave 0
            float L = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 8; ++w2)
                if (w2 < nw) L += st8[w2].y * __builtin_amdgcn_exp2f(st8[w2].x - M);
            const float f = __builtin_amdgcn_exp2f(mx - M) / L;
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                if (j < cnt) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][e] = fmaf(s[j][e], f, acc[j][e]);
                }
        }
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

    if (p.surgery) {
        run_sweep(0, 3 * H);
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
                    bf16x8 hi, lo;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        hi[k] = (__bf16)v[k];
                        lo[k] = (__bf16)(v[k] - (float)hi[k]);
                    }
                    u16* o = p.a_sum + ((long long)b * N + qg) * 2 * p.KP + tile * 64 + g8 * 8;
                    *reinterpret_cast<bf16x8*>(o) = hi;
                    *reinterpret_cast<bf16x8*>(o + 32) = lo;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }

    if (p.w_aff || p.attn_out) {
        run_sweep(3 * H, 4 * H);
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
int excel_launch_attn_strip(const unsigned short* qkvs, unsigned short* a_sum, float* w_aff, float* attn_out, int B, int H, int N,
                            int KP, int hd, float scale, int surgery, float w_scale, float aff_scale, int aff_init, const float* ex_attn,
                            hipStream_t st) {
    EXCEL_CHECK_ARG(hd == 64, "attention: head_dim must be 64 (got %d)", hd);
    EXCEL_CHECK_ARG(qkvs && (!surgery || (a_sum && KP == cdiv(N, 32) * 32)), "attn_strip: bad a_sum/KP");
    const int ntiles = cdiv(N, 32);
    if (ntiles > 40) return 1;                                  // > 8 waves x 5 tiles: not resident, use the two-pass kernels
    ProfScope prof__(PROF_ATTN_ACCUM, st);
    int ntw = cdiv(ntiles, 8);
    // prefer 7 waves when that keeps the per-SIMD tile load as even (25 tiles: 4,4,4,4,3,3,3)
    int nw = cdiv(ntiles, ntw);
    StripArgs a{qkvs, a_sum, w_aff, attn_out, surgery ? ex_attn : nullptr, B, H, N, KP, ntiles, cdiv(N, 32), scale, w_scale, aff_scale, (float)H,
                aff_init, surgery};
    const dim3 grid(B * a.nstrips), block(nw * 64);
    switch (ntw) {
        case 1: hipLaunchKernelGGL((attn_strip_kernel<1>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((attn_strip_kernel<2>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((attn_strip_kernel<3>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((attn_strip_kernel<4>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((attn_strip_kernel<5>), grid, block, 0, st, a); break;
    }
    EXCEL_CHECK_LAUNCH("attn_strip");
    return EXCEL_OK;
}
