// Self-attention kernels of the ExCEL "surgery" ViT (clip/clip_surgery_model.py:95-159, :307), fp32 on the
// f32-input matrix core so the softmaxes see exactly-fp32 scores.
//
// Data layout: the QKV GEMM writes q|k|v head-major, qkvh[B][3][H][N][64], so that a (b, type, head) matrix is
// one contiguous [N,64] slab and every 32/64-row tile is a single contiguous 8/16 KB read.
//
// Two kernels, both computing TRANSPOSED score tiles  S^T[key][q] = Y[key,:] . X[q,:]  so that a query row lives
// in ONE lane (q = lane & 31): row max / row sum are in-lane reductions plus one cross-half shuffle, and the
// probabilities feed the P.V product straight from their accumulator registers (no LDS round trip):
//
//   attn_rowpass  one pass over the keys per (b, head, type):
//                 type 0 (q.k): flash-style online softmax + O^T = V^T P^T  -> attention output, plus row stats
//                 type 1..3 (q.q, k.k, v.v; surgery blocks only): row stats (max, 1/sum) only
//   attn_accum    recomputes the score tiles with the final row stats and reduces over heads IN REGISTERS:
//                 A_sum = sum_h (softmax(qq)+softmax(kk)+softmax(vv))/3      (:125,:146)   [surgery blocks]
//                 W     = sum_h softmax(qk) (head-sum, :154) or head-mean (nn.MultiheadAttention, block 6)
//                 and folds W[1:,1:]/6 into the layer-mean affinity the random walk consumes (utils/affutils.py:180,197).
//   The N x N x heads x 4 probability tensors the reference materialises (118 MB/image/layer) never exist;
//   the price is one extra score GEMM per type, deterministic (no atomics).
#include <stdlib.h>
#include "common.h"
#include "excel_internal.h"

namespace EXCEL_SPLIT_NS {     // compiled once per 16-bit split type (excel_internal.h, build.py)

#define HD 64
#define KP 68   // LDS pitch (floats) of a [rows][64] operand tile read with ds_read_b128: slot = 17*row mod 16 -> conflict-free

// ------------------------------------------------------------------------------------------------ rowpass
struct RowpassArgs {
    const float* qkvh;   // [B,3,H,N,64]
    float* out;          // [B,N,H*64] attention output of type 0 (pre out-proj)
    float2* stats;       // [B,H,4,N] {row max (scaled scores), 1/row sum}
    int B, H, N;
    float scale;
    int out_split;       // 1: out is a split-bf16 tensor [B*N][2][H*64] (A operand of the bf16x3 out-proj GEMM)
    const unsigned short* qkvs;   // bf16x3 scores: q|k|v head-major in split format [B,3,H,N][2][64] (null = exact fp32 scores)
    int flash_nq;        // q-blocks >= flash_nq of type 0 only produce row stats (last block: only the cls row's output is consumed)
    const unsigned short* vt;     // (rounds 1-2: V^T for the bf16x3 P.V; the bf16x3 row pass reads V from qkvs through the LDS transpose read)
    int vt_kp;
    int xcd_local;       // 1: workgroups of one (image, head) on one XCD
    int tail_grp;        // (image, head) groups per chunk of the XCD-local order: their full q-blocks, then their partial last q-blocks
};

typedef unsigned short u16;

// BF = true: the score products run as bf16x3 (3 x v_mfma_f32_32x32x16_bf16 on split-bf16 q/k/v, see gemm_bf16x3.hip);
// the softmax and the P.V product stay fp32.  Score tiles keep the same accumulator layout, so everything downstream
// of the MFMAs is shared with the exact-fp32 path.
template <bool FLASH, bool BF>
__device__ __forceinline__ void rowpass_body(const RowpassArgs& p, float* smem, int b, int h, int type, int qblk) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, kh = lane >> 5;
    const int N = p.N;
    // type -> (row operand X, column operand Y) among q=0,k=1,v=2
    const int tx = (type == 0 || type == 1) ? 0 : (type == 2 ? 1 : 2);
    const int ty = (type == 0) ? 1 : tx;
    const float* X = p.qkvh + (((long long)b * 3 + tx) * p.H + h) * (long long)N * HD;
    const float* Y = p.qkvh + (((long long)b * 3 + ty) * p.H + h) * (long long)N * HD;
    const float* V = p.qkvh + (((long long)b * 3 + 2) * p.H + h) * (long long)N * HD;

    float* Ks = smem;                    // [2][32*KP]
    float* Vs = smem + 2 * 32 * KP;      // [2][32*64] fp32 V tile, or (PVBF) [2][64 d][32 hi | 32 lo] bf16 V^T tile
    // PVBF: the P.V product also runs as bf16x3.  The V^T tile row d holds this key tile's 32 keys as [hi 32 | lo 32]
    // bf16 = 16 eight-byte slots; slot (2c+half) is stored at (2c+half) ^ ((d>>1)&15): conflict-free ds_read_b64.
    const bool PVBF = FLASH && BF && p.vt != nullptr;
    const u16* VT = PVBF ? p.vt + (((long long)b * p.H + h) * 64) * 2 * p.vt_kp : nullptr;

    const int q0 = qblk * 128 + wave * 32;
    const int qrow = min(q0 + r, N - 1);
    f32x4 xf[8];
    splitx8 xh[4], xl[4];
    const u16* Ysp = nullptr;
    if (BF) {
        const u16* Xsp = p.qkvs + (((long long)b * 3 + tx) * p.H + h) * (long long)N * 128;
        Ysp = p.qkvs + (((long long)b * 3 + ty) * p.H + h) * (long long)N * 128;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            xh[s4] = *reinterpret_cast<const splitx8*>(Xsp + (long long)qrow * 128 + s4 * 16 + kh * 8);
            xl[s4] = *reinterpret_cast<const splitx8*>(Xsp + (long long)qrow * 128 + 64 + s4 * 16 + kh * 8);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) xf[c] = *reinterpret_cast<const f32x4*>(X + (long long)qrow * HD + c * 8 + kh * 4);
    }

    float m = -INFINITY, l = 0.f;
    f32x16 oT[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { oT[0][e] = 0.f; oT[1][e] = 0.f; }

    // staging map: a 32x64 tile = 512 float4, 2 per thread
    f32x4 rk[2], rv[2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int row = min(kt * 32 + (idx >> 4), N - 1), c4 = idx & 15;
            if (BF) rk[i] = *reinterpret_cast<const f32x4*>(Ysp + (long long)row * 128 + c4 * 8);   // 16-B chunk c4 of [hi 64 | lo 64]
            else rk[i] = *reinterpret_cast<const f32x4*>(Y + (long long)row * HD + c4 * 4);
            if (FLASH) {
                if (PVBF) {   // chunk (idx & 7) of d-row (idx >> 3): 16 B of [hi 32 | lo 32] of key block kt
                    const int d = idx >> 3, c = idx & 7;
                    rv[i] = *reinterpret_cast<const f32x4*>(VT + (long long)d * 2 * p.vt_kp + kt * 64 + c * 8);
                } else {
                    rv[i] = *reinterpret_cast<const f32x4*>(V + (long long)row * HD + c4 * 4);
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, c4 = idx & 15;
            if (BF)   // 256-B rows of 16 chunks, chunk c at slot c ^ (row & 15): conflict-free ds_read_b128 without padding
                *reinterpret_cast<f32x4*>(reinterpret_cast<u16*>(Ks) + buf * 32 * 128 + row * 128 + ((c4 ^ (row & 15)) * 8)) = rk[i];
            else
                *reinterpret_cast<f32x4*>(&Ks[buf * 32 * KP + row * KP + c4 * 4]) = rk[i];
            if (FLASH) {
                if (PVBF) {
                    const int d = idx >> 3, c = idx & 7, msk = (d >> 1) & 15;
                    f32x4 v = rv[i];
                    if (msk & 1) v = f32x4{v[2], v[3], v[0], v[1]};      // the two 8-B halves of the chunk swap slots
                    *reinterpret_cast<f32x4*>(reinterpret_cast<u16*>(Vs) + buf * 64 * 64 + d * 64 + ((c ^ (msk >> 1)) * 8)) = v;
                } else {
                    *reinterpret_cast<f32x4*>(&Vs[buf * 32 * 64 + row * 64 + c4 * 4]) = rv[i];
                }
            }
        }
    };

    const int nkt = (N + 31) / 32;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load_tile(kt + 1);
        const float* ks = Ks + cur * 32 * KP;
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        if (BF) {
            const u16* kr = reinterpret_cast<const u16*>(Ks) + cur * 32 * 128 + r * 128;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const splitx8 yh = *reinterpret_cast<const splitx8*>(kr + (((s4 * 2 + kh) ^ (r & 15)) * 8));
                const splitx8 yl = *reinterpret_cast<const splitx8*>(kr + (((8 + s4 * 2 + kh) ^ (r & 15)) * 8));
                s = EXCEL_MFMA16(yl, xh[s4], s, 0, 0, 0);
                s = EXCEL_MFMA16(yh, xl[s4], s, 0, 0, 0);
                s = EXCEL_MFMA16(yh, xh[s4], s, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 yf = *reinterpret_cast<const f32x4*>(&ks[r * KP + c * 8 + kh * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(yf[e], xf[c][e], s, 0, 0, 0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = kt * 32 + c32_row(e, lane);
            s[e] = (key < N) ? s[e] * p.scale : -INFINITY;
            mx = fmaxf(mx, s[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        const float alpha = __expf(m - m_new);
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = __expf(s[e] - m_new);
            ps += s[e];
        }
        ps += __shfl_xor(ps, 32, 64);
        l = l * alpha + ps;
        m = m_new;
        if (FLASH) {
            const float* vs = Vs + cur * 32 * 64;
#pragma unroll
            for (int e = 0; e < 16; ++e) { oT[0][e] *= alpha; oT[1][e] *= alpha; }
            if (PVBF) {
                // P (this lane: keys (e&3)+8(e>>2)+4kh of query r) -> bf16 hi/lo; MFMA k-step ks takes e = 8ks..8ks+7, i.e. keys
                // {16ks+4kh+0..3, 16ks+8+4kh+0..3}: the V^T operand reads exactly those two 8-byte groups of row d.
                splitx8 ph[2], pl[2];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const split_t hi = split_hi(s[e]);
                    ph[e >> 3][e & 7] = hi;
                    pl[e >> 3][e & 7] = split_hi(s[e] - (float)hi);
                }
                const u16* vt16 = reinterpret_cast<const u16*>(Vs) + cur * 64 * 64;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int d = dt * 32 + r, msk = (d >> 1) & 15;
                    const u16* rowp = vt16 + d * 64;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        // 8-byte slot index of (chunk c, half kh) = 2c + kh; hi chunks 0..3, lo chunks 4..7
                        const splitx4 h0 = *reinterpret_cast<const splitx4*>(rowp + (((2 * (2 * ks) + kh) ^ msk) * 4));
                        const splitx4 h1 = *reinterpret_cast<const splitx4*>(rowp + (((2 * (2 * ks + 1) + kh) ^ msk) * 4));
                        const splitx4 l0 = *reinterpret_cast<const splitx4*>(rowp + (((2 * (4 + 2 * ks) + kh) ^ msk) * 4));
                        const splitx4 l1 = *reinterpret_cast<const splitx4*>(rowp + (((2 * (5 + 2 * ks) + kh) ^ msk) * 4));
                        const splitx8 vh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                        const splitx8 vl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                        oT[dt] = EXCEL_MFMA16(vl, ph[ks], oT[dt], 0, 0, 0);
                        oT[dt] = EXCEL_MFMA16(vh, pl[ks], oT[dt], 0, 0, 0);
                        oT[dt] = EXCEL_MFMA16(vh, ph[ks], oT[dt], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int krow = c32_row(e, lane);
                    const float v0 = vs[krow * 64 + r];
                    const float v1 = vs[krow * 64 + 32 + r];
                    oT[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[e], oT[0], 0, 0, 0);
                    oT[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[e], oT[1], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    const float linv = 1.f / l;
    if (kh == 0 && q0 + r < N)
        p.stats[(((long long)b * p.H + h) * 4 + type) * N + q0 + r] = make_float2(m, linv);

    if (FLASH) {
        // O^T (d spread over registers, q per lane) -> LDS [q][d] (pitch 65) -> coalesced 256-B row stores
        float* ob = smem + wave * (32 * 65);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) ob[r * 65 + dt * 32 + c32_row(e, lane)] = oT[dt][e] * linv;
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes landed (same-wave readback)
        for (int qq = 0; qq < 32; ++qq) {
            const int q = q0 + qq;
            if (q >= N) break;
            const float v = ob[qq * 65 + lane];
            if (p.out_split) {
                const split_t hi = split_hi(v);
                split_t* o = reinterpret_cast<split_t*>(p.out) + ((long long)b * N + q) * 2 * (p.H * HD) + split_off(h * HD + lane, 0);
                o[0] = hi;
                o[32] = split_hi(v - (float)hi);
            } else {
                p.out[((long long)b * N + q) * (p.H * HD) + h * HD + lane] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ rowpass, bf16x3 pipeline
// Production variant of the row pass (bf16x3 scores AND bf16x3 P.V): K (split q|k|v rows) and V^T tiles stream through a
// 3-stage LDS ring filled by global_load_lds (no VGPR round trip), two key tiles in flight behind a counted s_waitcnt
// vmcnt + raw s_barrier, exactly like gemm_bf16x3_kernel<4,3>: the register-staged version waited for each tile's global
// loads inside the step that issued them.  LDS images are lane-linear, so both swizzles are applied to the SOURCE address:
//   K tile   [32 keys][16 chunks of 16 B = hi 64 | lo 64]   chunk c at slot c ^ (key & 15)      (conflict-free b128)
//   V tile   8 sub-tiles (hi d 0-15, 16-31, 32-47, 48-63, then lo) of [32 keys][16 d] bf16, 1 KB each, 1152 B apart: the P.V operand
//            (lane = d, 4 consecutive keys) is read with ds_read_b64_tr_b16 - the hardware transposes a [4 keys][16 d] block per 16
//            lanes, each lane supplying the address of one 8-byte row segment - straight from the row-major q|k|v planes.  Rounds 1-3
//            read it from a V^T copy that a separate transpose kernel wrote per block (0.39 ms per step, 2-way LDS conflicts).
template <bool FLASH>
__device__ __forceinline__ void rowpass_body_bf(const RowpassArgs& p, float* smem, int b, int h, int type, int qblk) {
    constexpr int KT_EL = 32 * 128;                      // u16 elements of a K tile (8 KB)
    constexpr int VSUB = 1152;                           // bytes between V sub-tiles: 1 KB + 128 (the two sub-tiles a half-wave reads sit in complementary banks)
    constexpr int VT_EL = 8 * VSUB / 2;                  // u16 elements of a V tile (9 KB)
    constexpr int STAGE_EL = KT_EL + (FLASH ? VT_EL : 0);
    constexpr int PER_WAVE = FLASH ? 4 : 2;              // 1-KB global_load_lds per wave per key tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kh = lane >> 5;
    const int N = p.N;
    const int tx = (type == 0 || type == 1) ? 0 : (type == 2 ? 1 : 2);
    const int ty = (type == 0) ? 1 : tx;
    const u16* Xsp = p.qkvs + (((long long)b * 3 + tx) * p.H + h) * (long long)N * 128;
    const u16* Ysp = p.qkvs + (((long long)b * 3 + ty) * p.H + h) * (long long)N * 128;
    const u16* Vsp = p.qkvs + (((long long)b * 3 + 2) * p.H + h) * (long long)N * 128;
    u16* ring = reinterpret_cast<u16*>(smem);            // [3][K tile | V^T tile]
    const unsigned ring_b = __builtin_amdgcn_readfirstlane(lds_addr(ring));      // (provably wave-uniform: it goes into m0)

    const int q0 = qblk * 128 + wave * 32;
    const int qrow = min(q0 + r, N - 1);
    splitx8 xh[4], xl[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        xh[s4] = *reinterpret_cast<const splitx8*>(Xsp + (long long)qrow * 128 + s4 * 16 + kh * 8);
        xl[s4] = *reinterpret_cast<const splitx8*>(Xsp + (long long)qrow * 128 + 64 + s4 * 16 + kh * 8);
    }

    // consume the query fragments once BEFORE any LDS-DMA is in flight: the compiler then places its wait for these ordinary loads
    // here and not (as vmcnt(0), draining the DMA queue) in front of their first use inside the key loop
    asm volatile("" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[3]));
    // LDS-DMA through buffer descriptors over this (image, head)'s key plane and V^T rows: the per-lane byte offsets are loop invariant
    // (2 + 2 VGPRs), the tile position goes into the scalar offset - no per-lane 64-bit address arithmetic and no row clamp inside the key
    // loop (it was ~14 VALU instructions per key tile in a loop that is bound by instruction issue).  Key rows past the end of the plane
    // read the next plane / the descriptor's out-of-range zeros: finite, and masked to -inf below; V^T is zero padded to vt_kp keys.
    typedef __attribute__((address_space(3))) unsigned char* lds_bptr;
    const long long kplane_bytes = (long long)N * 256;
    const long long kavail = ((((long long)p.B * 3 - ((long long)b * 3 + ty)) * p.H - h) * kplane_bytes);      // bytes from this plane to the end of qkvs
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)Ysp, 0, (int)(kavail < (1LL << 31) - 1 ? kavail : (1LL << 31) - 1), 0x00020000);
    const long long vavail = ((((long long)p.B * 3 - ((long long)b * 3 + 2)) * p.H - h) * kplane_bytes);       // ... from the v plane to the end of qkvs
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)Vsp, 0, FLASH ? (int)(vavail < (1LL << 31) - 1 ? vavail : (1LL << 31) - 1) : 0, 0x00020000);
    int koffb[2], voffb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int krow = wave * 8 + j * 4 + (lane >> 4);                 // row of the tile this lane loads a chunk of
        koffb[j] = krow * 256 + (((lane & 15) ^ (krow & 15)) * 16);
        const int sub = wave * 2 + j;                                    // V sub-tile this lane loads 16 B of: key lane >> 1, d half lane & 1
        voffb[j] = (lane >> 1) * 256 + (sub >> 2) * 128 + (((sub & 3) * 16 + (lane & 1) * 8) * 2);
    }
    // piece idx of this wave's PER_WAVE 1-KB loads of key tile kt: 0, 1 = K rows, 2, 3 = V^T rows (flash only)
    auto issue_piece = [&](int kt, int stage, int idx) {
        unsigned dstk = __builtin_amdgcn_readfirstlane(ring_b + stage * (STAGE_EL * 2));     // wave-uniform (it goes into m0); also keeps
        asm volatile("" : "+s"(dstk));                                                         // the per-piece destinations out of hoisted SGPRs
        if (idx < 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_bptr)(unsigned long long)(dstk + (wave * 8 + idx * 4) * 256), 16, koffb[idx], kt * (32 * 256), 0, 0);
        } else if (FLASH) {
            const int j = idx - 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_bptr)(unsigned long long)(dstk + KT_EL * 2 + (wave * 2 + j) * VSUB), 16, voffb[j], kt * (32 * 256), 0, 0);
        }
    };
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int idx = 0; idx < PER_WAVE; ++idx) issue_piece(kt, stage, idx);
    };

    const float c2 = p.scale * 1.4426950408889634f;     // scores enter the softmax in log2 units
    float m = -INFINITY, l = 0.f;                         // m: running row max in log2 units
    f32x16 oT[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { oT[0][e] = 0.f; oT[1][e] = 0.f; }

    const int nkt = (N + 31) / 32;
    issue(0, 0);
    if (nkt > 1) issue(1, 1);
    int stage = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) {
            if (PER_WAVE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const int rstage = stage == 0 ? 2 : stage - 1;                       // slot of tile kt-1, free since the barrier: tile kt+2 goes there
        const bool refill = kt + 2 < nkt;
        if (q0 >= N) {                                                        // wave-uniform: all 32 query rows are padding -> only keep the ring moving
            if (refill) issue(kt + 2, rstage);
            stage = (stage == 2) ? 0 : stage + 1;
            continue;
        }
        // LDS reads as inline asm (common.h): a compiler-visible ds_read behind pending LDS-DMA gets an s_waitcnt vmcnt(0) in front of
        // it, which drained the two key tiles in flight on every step
        const unsigned kr = ring_b + (stage * STAGE_EL + r * 128) * 2;
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        {
            splitx8 yh[4], yl[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                yh[s4] = lds_read16(kr + (((s4 * 2 + kh) ^ (r & 15)) * 16));
                yl[s4] = lds_read16(kr + (((8 + s4 * 2 + kh) ^ (r & 15)) * 16));
            }
            lds_wait8(yh, yl);
            // the refill is issued one piece per k-step between the score MFMAs (back to back, the pieces of all waves queue up in the CU's
            // one address path and each wave sits behind its own before it reaches its MFMAs)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if (refill && s4 < PER_WAVE) issue_piece(kt + 2, rstage, s4);
                s = EXCEL_MFMA16(yl[s4], xh[s4], s, 0, 0, 0);
                s = EXCEL_MFMA16(yh[s4], xl[s4], s, 0, 0, 0);
                s = EXCEL_MFMA16(yh[s4], xh[s4], s, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // softmax bookkeeping in log2 units: p = exp2(s*c2 - m2), c2 = scale*log2(e)  (one fma + one v_exp per element);
        // only the last key tile can contain keys >= N, so the mask lives in a uniform branch.
        if (kt == nkt - 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (kt * 32 + c32_row(e, lane) >= N) s[e] = -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
        {
            float mlo, mhi;
            wave_halves(mx, mlo, mhi);
            mx = fmaxf(mlo, mhi) * c2;
        }
        const float m_new = fmaxf(m, mx);
        const bool grew = m_new > m;
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);            // 1.0 exactly when the running max did not move
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = __builtin_amdgcn_exp2f(fmaf(s[e], c2, -m_new));
            ps += s[e];
        }
        {
            float plo, phi;
            wave_halves(ps, plo, phi);
            ps = plo + phi;
        }
        l = l * alpha + ps;
        m = m_new;
        if (FLASH) {
            if (__any(grew)) {                           // wave-uniform: after the first few tiles the max rarely moves
#pragma unroll
                for (int e = 0; e < 16; ++e) { oT[0][e] *= alpha; oT[1][e] *= alpha; }
            }
            // P (this lane: keys (e&3)+8(e>>2)+4kh of query r) -> bf16 hi/lo; MFMA k-step ks takes e = 8ks..8ks+7, i.e. keys
            // {16ks+4kh+0..3, 16ks+8+4kh+0..3}: the V operand (lane = d) takes exactly those two groups of 4 consecutive keys.
            // hi = p truncated to bf16 (bit mask), lo = bf16(p - hi): p - hi is exact, so hi + lo keeps 16 mantissa bits.
            splitx8 ph[2], pl[2];
#ifdef EXCEL_SPLIT_F16
            {   // IEEE half: probabilities need neither the saturation nor the NaN test of split_hi (common.h: split_pair_bounded)
                unsigned phu[2][4], plu[2][4];
#pragma unroll
                for (int e = 0; e < 16; e += 2) split_pair_bounded(s[e], s[e + 1], phu[e >> 3][(e & 7) >> 1], plu[e >> 3][(e & 7) >> 1]);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    ph[q] = __builtin_bit_cast(splitx8, uint4{phu[q][0], phu[q][1], phu[q][2], phu[q][3]});
                    pl[q] = __builtin_bit_cast(splitx8, uint4{plu[q][0], plu[q][1], plu[q][2], plu[q][3]});
                }
            }
#else
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float hf = __uint_as_float(__float_as_uint(s[e]) & 0xFFFF0000u);
                ph[e >> 3][e & 7] = split_hi(hf);
                pl[e >> 3][e & 7] = split_hi(s[e] - hf);
            }
#endif
            // lane (d = 32 dt + r, kh): sub-tile 2 dt + (r >> 4), its column r & 15; the 16 lanes of a group address the 16 row segments
            // (key = k0 + i / 4, d quarter i % 4) of the [4 keys][16 d] block whose column they receive
            const unsigned vb = ring_b + (stage * STAGE_EL + KT_EL) * 2 + ((lane >> 4) & 1) * VSUB + (4 * kh + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                splitx4 vq[8];                                   // [ks][h0 h1 l0 l1]
                const unsigned vbd = vb + dt * 2 * VSUB;        // (one address per d tile; everything else is the instruction's immediate)
                vq[0] = lds_read8h_tr<0>(vbd);
                vq[1] = lds_read8h_tr<256>(vbd);
                vq[2] = lds_read8h_tr<4 * VSUB>(vbd);
                vq[3] = lds_read8h_tr<4 * VSUB + 256>(vbd);
                vq[4] = lds_read8h_tr<512>(vbd);
                vq[5] = lds_read8h_tr<512 + 256>(vbd);
                vq[6] = lds_read8h_tr<4 * VSUB + 512>(vbd);
                vq[7] = lds_read8h_tr<4 * VSUB + 512 + 256>(vbd);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vq[0]), "+v"(vq[1]), "+v"(vq[2]), "+v"(vq[3]), "+v"(vq[4]), "+v"(vq[5]), "+v"(vq[6]), "+v"(vq[7])::"memory");
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const splitx4 h0 = vq[ks * 4], h1 = vq[ks * 4 + 1], l0 = vq[ks * 4 + 2], l1 = vq[ks * 4 + 3];
                    const splitx8 vh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const splitx8 vl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    oT[dt] = EXCEL_MFMA16(vl, ph[ks], oT[dt], 0, 0, 0);
                    oT[dt] = EXCEL_MFMA16(vh, pl[ks], oT[dt], 0, 0, 0);
                    oT[dt] = EXCEL_MFMA16(vh, ph[ks], oT[dt], 0, 0, 0);
                }
            }
        }
        stage = (stage == 2) ? 0 : stage + 1;
    }

    const float linv = 1.f / l;
    if (kh == 0 && q0 + r < N)
        p.stats[(((long long)b * p.H + h) * 4 + type) * N + q0 + r] = make_float2(m, linv);   // m in LOG2 units (bf16x3 accum expects that)

    if (FLASH) {
        __syncthreads();     // every wave has finished reading the ring: reuse it for the output transpose
        float* ob = smem + wave * (32 * 65);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) ob[r * 65 + dt * 32 + c32_row(e, lane)] = oT[dt][e] * linv;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int qq = 0; qq < 32; ++qq) {
            const int q = q0 + qq;
            if (q >= N) break;
            const float v = ob[qq * 65 + lane];
            if (p.out_split) {
                const split_t hi = split_hi(v);
                split_t* o = reinterpret_cast<split_t*>(p.out) + ((long long)b * N + q) * 2 * (p.H * HD) + split_off(h * HD + lane, 0);
                o[0] = hi;
                o[32] = split_hi(v - (float)hi);
            } else {
                p.out[((long long)b * N + q) * (p.H * HD) + h * HD + lane] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void attn_rowpass_kernel(RowpassArgs p) {
    __shared__ __attribute__((aligned(1024))) float smem[3 * (2048 + 2304)];   // 51 KB: 3-stage ring of (K tile 8 KB + V tile 9 KB); >= the 33.8 KB of the fp32 path
    int bh = blockIdx.y, qb = blockIdx.x;
    if (p.xcd_local) {
        // the q-blocks of one (image, head) share its K / V^T tiles: keep them on one XCD (consecutive logical ids) so the tiles are
        // fetched from the fabric once, not once per XCD (measured fabric traffic of this kernel: 2.6x its algorithmic bytes)
        const int nq = gridDim.x, nbh = gridDim.y, lin = blockIdx.x + nq * blockIdx.y;
        if ((nbh & 7) == 0 && nq > 1 && (p.N & 127) != 0) {
            // ... and, inside an XCD's chunk, the full q-blocks first and the partial last q-block of every (image, head) at the end:
            // 384 x 7 workgroups on 768 slots are 3.5 rounds; 384 x 6 full ones are exactly 3, and the tail round is then made of the
            // short blocks (17 of 128 rows at N = 785: one active wave) instead of a half-empty round of full ones
            // Round 4: not ONE tail per XCD but one per chunk of `tail_grp` groups: the partial block of a group then runs while the
            // group's K / V tiles (400 KB per (image, head)) are still in the XCD's 4-MB L2 - at the very end every partial block
            // re-fetched them from the fabric (PMC: 463 MB per launch against 312 MB before the tail order, ~308 MB algorithmic).
            const int x = lin & 7, loc = lin >> 3, per = nbh >> 3;
            const int grp = min(max(p.tail_grp, 1), per), cs = grp * nq;
            const int c = loc / cs, within = loc - c * cs;
            const int g0 = c * grp, gcount = min(grp, per - g0), nfull = gcount * (nq - 1);
            if (within < nfull) { bh = x * per + g0 + within / (nq - 1); qb = within % (nq - 1); }
            else { bh = x * per + g0 + (within - nfull); qb = nq - 1; }
        } else {
            const int id = xcd_remap(lin, nq * nbh);
            qb = id % nq;
            bh = id / nq;
        }
    }
    const int b = bh / p.H, h = bh % p.H;
    const int type = blockIdx.z;
    const bool flash = type == 0 && qb < p.flash_nq;
    if (p.qkvs) {
        if (flash) rowpass_body_bf<true>(p, smem, b, h, 0, qb);
        else rowpass_body_bf<false>(p, smem, b, h, type, qb);
    } else {
        if (flash) rowpass_body<true, false>(p, smem, b, h, 0, qb);
        else rowpass_body<false, false>(p, smem, b, h, type, qb);
    }
}

// ------------------------------------------------------------------------------------------------ accum
struct AccumArgs {
    const float* qkvh;    // [B,3,H,N,64]
    const float2* stats;  // [B,H,4,N]
    float* a_sum;         // [B,N,NP]  (surgery only) head-sum of (qq+kk+vv softmaxes)/3, zero in columns [N,NP)
    float* w_aff;         // [B,P,P]   running layer-mean of W[1:,1:]   (may be null)
    float* attn_out;      // [B,N,N]   W of this layer (may be null)
    int B, H, N, NP;
    float scale;
    float w_scale;        // 1/H for nn.MultiheadAttention blocks (head-mean), 1 for surgery blocks (head-sum)
    float aff_scale;      // 1/attn_layers
    int aff_init;         // 1: w_aff = ..., 0: w_aff += ...
    const unsigned short* qkvs;   // split-bf16 q|k|v for bf16x3 scores (null = exact fp32)
    int dbg;              // dev: bit0 skip scoring, bit1 skip tile loads, bit2 skip the output epilogue
    int a_sum_split;      // 1: a_sum is written in split-bf16 format [B,N][2*NP] (NP % 32 == 0): A operand of the bf16x3 A_sum.V GEMM
    const float* ex_attn; // [B,P,P] LVC cue added to every head's attn[1:,1:] of a surgery block (may be null)
    float ex_scale;       // = H (the head sum of a per-head constant)
};

template <bool SURGERY, bool BF>
__global__ __launch_bounds__(256, 1) void attn_accum_kernel(AccumArgs p) {
    constexpr int NT = SURGERY ? 6 : 2;
    __shared__ __attribute__((aligned(16))) float tiles[NT * 64 * KP];   // 104,448 B / 34,816 B (BF uses 64*64 floats per tile)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, kh = lane >> 5;
    const int wk = wave >> 1, wq = wave & 1;
    const int kt = blockIdx.x, qt = blockIdx.y, b = blockIdx.z;
    const int N = p.N;
    const int q = qt * 64 + wq * 32 + r;
    const int qc = min(q, N - 1);

    f32x16 accW, accA;
#pragma unroll
    for (int e = 0; e < 16; ++e) { accW[e] = 0.f; accA[e] = 0.f; }

    // tile slots: X (query-side rows qt*64..) = q,k,v -> 0,1,2 ; Y (key-side rows kt*64..) = q,k,v -> 3,4,5
    // non-surgery: slot 0 = X q, slot 1 = Y k
    for (int h = 0; h < p.H; ++h) {
        __syncthreads();   // previous head's fragment reads done
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int typ, row0;
            if (SURGERY) { typ = t % 3; row0 = (t < 3) ? qt * 64 : kt * 64; }
            else { typ = t; row0 = (t == 0) ? qt * 64 : kt * 64; }
            const float* src = p.qkvh + (((long long)b * 3 + typ) * p.H + h) * (long long)N * HD;
            const u16* srcs = BF ? p.qkvs + (((long long)b * 3 + typ) * p.H + h) * (long long)N * 128 : nullptr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i;
                const int row = idx >> 4, c4 = idx & 15;
                if (BF) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(srcs + (long long)min(row0 + row, N - 1) * 128 + c4 * 8);
                    *reinterpret_cast<f32x4*>(reinterpret_cast<u16*>(tiles) + t * 64 * 128 + row * 128 + ((c4 ^ (row & 15)) * 8)) = v;
                } else {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long long)min(row0 + row, N - 1) * HD + c4 * 4);
                    *reinterpret_cast<f32x4*>(&tiles[t * 64 * KP + row * KP + c4 * 4]) = v;
                }
            }
        }
        __syncthreads();

        const float2* st = p.stats + ((long long)b * p.H + h) * 4 * N;
        auto score = [&](int slotY, int slotX, int type, f32x16& acc) {
            const float2 ml = st[(long long)type * N + qc];
            const float* ys = tiles + slotY * 64 * KP + (wk * 32 + r) * KP + kh * 4;
            const float* xs = tiles + slotX * 64 * KP + (wq * 32 + r) * KP + kh * 4;
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            if (BF) {
                const u16* y16 = reinterpret_cast<const u16*>(tiles) + slotY * 64 * 128 + (wk * 32 + r) * 128;
                const u16* x16 = reinterpret_cast<const u16*>(tiles) + slotX * 64 * 128 + (wq * 32 + r) * 128;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int ch = ((s4 * 2 + kh) ^ (r & 15)) * 8, cl = ((8 + s4 * 2 + kh) ^ (r & 15)) * 8;
                    const splitx8 yh = *reinterpret_cast<const splitx8*>(y16 + ch), yl = *reinterpret_cast<const splitx8*>(y16 + cl);
                    const splitx8 xh = *reinterpret_cast<const splitx8*>(x16 + ch), xl = *reinterpret_cast<const splitx8*>(x16 + cl);
                    s = EXCEL_MFMA16(yl, xh, s, 0, 0, 0);
                    s = EXCEL_MFMA16(yh, xl, s, 0, 0, 0);
                    s = EXCEL_MFMA16(yh, xh, s, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x4 yf = *reinterpret_cast<const f32x4*>(ys + c * 8);
                    const f32x4 xf = *reinterpret_cast<const f32x4*>(xs + c * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(yf[e], xf[e], s, 0, 0, 0);
                }
            }
            if (BF) {
                // bf16x3 path: row stats were written in log2 units by rowpass_body_bf
                const float c2 = p.scale * 1.4426950408889634f;
                if (kt * 64 + wk * 32 + 32 > N) {       // wave-uniform: only the last key tile holds keys >= N
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kt * 64 + wk * 32 + c32_row(e, lane);
                        const float pr = __builtin_amdgcn_exp2f(fmaf(s[e], c2, -ml.x));
                        acc[e] += (key < N) ? pr * ml.y : 0.f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = fmaf(__builtin_amdgcn_exp2f(fmaf(s[e], c2, -ml.x)), ml.y, acc[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * 64 + wk * 32 + c32_row(e, lane);
                    const float pr = __expf(s[e] * p.scale - ml.x) * ml.y;
                    acc[e] += (key < N) ? pr : 0.f;
                }
            }
        };
        if (SURGERY) {
            score(4, 0, 0, accW);   // q.k
            score(3, 0, 1, accA);   // q.q
            score(4, 1, 2, accA);   // k.k
            score(5, 2, 3, accA);   // v.v
        } else {
            score(1, 0, 0, accW);
        }
    }
    __syncthreads();

    // transpose each wave's [key][q] tile through LDS (pitch 33) and store rows of q with consecutive keys
    float* tb = tiles + wave * (32 * 33);
    const int qbase = qt * 64 + wq * 32, kbase = kt * 64 + wk * 32;
    auto emit = [&](const f32x16& acc, int which) {
#pragma unroll
        for (int e = 0; e < 16; ++e) tb[r * 33 + c32_row(e, lane)] = acc[e];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // unrolled: the 16 read-modify-write round trips on w_aff must be in flight together, not one after the other
        // (a rolled loop serialised 16 dependent global loads per wave and dominated the kernel)
        float oldw[16];
        if (which == 1 && p.w_aff && !p.aff_init) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qg = qbase + 2 * i + kh, kg = kbase + r;
                const long long P = N - 1;
                oldw[i] = (qg < N && qg >= 1 && kg >= 1 && kg < N) ? p.w_aff[((long long)b * P + (qg - 1)) * P + (kg - 1)] : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) oldw[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qq = 2 * i + kh;
            const int qg = qbase + qq, kg = kbase + r;
            const float v = tb[qq * 33 + r];
            if (qg >= N) continue;
            if (which == 0) {
                if (kg < p.NP) {
                    float av = v * (1.f / 3.f);
                    // LVC branch (clip_surgery_model.py:140-141): every head's attn[1:,1:] += ex_attn -> head-sum gains H x ex_attn
                    if (p.ex_attn && qg >= 1 && kg >= 1 && kg < N) av += p.ex_scale * p.ex_attn[((long long)b * (N - 1) + (qg - 1)) * (N - 1) + (kg - 1)];
                    if (p.a_sum_split) {
                        split_t* o = reinterpret_cast<split_t*>(p.a_sum) + ((long long)b * N + qg) * 2 * p.NP + split_off(kg, 0);
                        const split_t hi = split_hi(av);
                        o[0] = hi;
                        o[32] = split_hi(av - (float)hi);
                    } else {
                        p.a_sum[((long long)b * N + qg) * p.NP + kg] = av;
                    }
                }
            } else {
                const float pw = v * p.w_scale;
                if (p.attn_out && kg < N) p.attn_out[((long long)b * N + qg) * N + kg] = pw;
                if (p.w_aff && qg >= 1 && kg >= 1 && kg < N) {
                    const long long P = N - 1;
                    float* dst = p.w_aff + ((long long)b * P + (qg - 1)) * P + (kg - 1);
                    *dst = oldw[i] + pw * p.aff_scale;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    };
    if (SURGERY) emit(accA, 0);
    emit(accW, 1);
}

// ------------------------------------------------------------------------------------------------ accum, bf16x3 pipeline
// Production variant of the accumulate pass.  Same maths as attn_accum_kernel<.,true>, restructured after measuring it:
//   * workgroup = 128 queries x 64 keys, 8 waves (4 x 2, one 32x32 score tile each): two waves per SIMD, so one wave's
//     exp/accumulate VALU work overlaps the other's MFMAs, and 25 % fewer operand bytes per score than 64 x 64;
//   * the six operand tiles (q|k|v rows of the query tile X: 32 KB each, of the key tile Y: 16 KB each) and the row stats
//     are streamed by global_load_lds on a ROLLING schedule, so the tiles of head h+1 land while head h is scored.
// Per head, 4 steps, one raw s_barrier each (X tile = 4, Y tile = 2, stats = 2 wave-instructions per wave):
//     step 0  qq (XQ,YQ)  issue XV,YV(h)                 wait XQ,YQ,stats(h): newest XK,YK(h)          -> vmcnt(6)
//     step 1  qk (XQ,YK)  issue YQ,stats(h+1)            wait YK(h):          newest XV,YV(h)          -> vmcnt(6)
//     step 2  kk (XK,YK)  issue XQ(h+1)                  (barrier only: frees XQ)
//     step 3  vv (XV,YV)  issue XK,YK(h+1)               wait XV,YV(h):       newest YQ,stats,XQ(h+1)  -> vmcnt(8) / (0) last
// A slot is refilled only after the barrier that follows its last reader; vmcnt retires loads in issue order.  There is
// no ordinary global load inside the loop (it would force a vmcnt(0) drain of the LDS-DMA queue).
template <bool SURGERY>
__global__ __launch_bounds__(512, 2) void attn_accum_bf_kernel(AccumArgs p) {
    constexpr int XT = 128 * 128, YT = 64 * 128;                    // u16 elements of an X tile (32 KB) / Y tile (16 KB)
    constexpr int NTYPE = SURGERY ? 4 : 1;
    // surgery: XQ XK XV | YQ YK YV ; else double-buffered (XQ, YK)
    constexpr int TILES_EL = SURGERY ? 3 * XT + 3 * YT : 2 * (XT + YT);
    __shared__ __attribute__((aligned(1024))) u16 tiles[TILES_EL];
    __shared__ __attribute__((aligned(1024))) float2 lstats[2 * NTYPE * 128 + 32];   // [buffer][type][128 q] + a dump slot
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kh = lane >> 5;
    const int wq = wave >> 1, wk = wave & 1;
    int kt = blockIdx.x, qt = blockIdx.y, b = blockIdx.z;
    if (EXCEL_DBG(p.dbg) & 8) {      // experiment: XCD-contiguous tile order (workgroups sharing X / Y tiles on one L2)
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int id = xcd_remap(lin, gridDim.x * gridDim.y * gridDim.z);
        kt = id % gridDim.x; qt = (id / gridDim.x) % gridDim.y; b = id / (gridDim.x * gridDim.y);
    }
    const int N = p.N;

    // staging: the tile image is lane-linear (one wave-instruction = 1 KB = 4 rows x 16 chunks), so the swizzle
    // (chunk c at slot c ^ (row & 15)) is applied to the source address.  X: wave w rows 16w..16w+15 (4 instr);
    // Y: rows 8w..8w+7 (2 instr).
    int rowx[4], coffx[4], rowy[2], coffy[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row_l = wave * 16 + j * 4 + (lane >> 4);
        coffx[j] = ((lane & 15) ^ (row_l & 15)) * 8;
        rowx[j] = min(qt * 128 + row_l, N - 1);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row_l = wave * 8 + j * 4 + (lane >> 4);
        coffy[j] = ((lane & 15) ^ (row_l & 15)) * 8;
        rowy[j] = min(kt * 64 + row_l, N - 1);
    }
    auto issue_x = [&](int off, int typ, int h) {
        const u16* src = p.qkvs + (((long long)b * 3 + typ) * p.H + h) * (long long)N * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long long)rowx[j] * 128 + coffx[j]),
                                             (__attribute__((address_space(3))) void*)(tiles + off + (wave * 16 + j * 4) * 128), 16, 0, 0);
    };
    auto issue_y = [&](int off, int typ, int h) {
        const u16* src = p.qkvs + (((long long)b * 3 + typ) * p.H + h) * (long long)N * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long long)rowy[j] * 128 + coffy[j]),
                                             (__attribute__((address_space(3))) void*)(tiles + off + (wave * 8 + j * 4) * 128), 16, 0, 0);
    };
    // row stats of head h: NTYPE x 128 float2 = NTYPE KB; every wave moves 2 x 256 B (global_load_lds_dword)
    auto issue_stats = [&](int h) {
        float* dst = reinterpret_cast<float*>(lstats + (h & 1) * NTYPE * 128);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave * 2 + j;                          // 16 pieces of 64 floats (= 32 float2)
            if (SURGERY || piece < 4) {
                const int ty = piece >> 2, q32 = (piece & 3) * 32 + (lane >> 1);
                const float* src = reinterpret_cast<const float*>(p.stats + (((long long)b * p.H + h) * 4 + ty) * N + min(qt * 128 + q32, N - 1)) + (lane & 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + piece * 64), 4, 0, 0);
            } else {   // keep the per-wave instruction count uniform (counted vmcnt): harmless reload of piece 0
                const float* src = reinterpret_cast<const float*>(p.stats + (((long long)b * p.H + h) * 4) * N + min(qt * 128 + (lane >> 1), N - 1)) + (lane & 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(reinterpret_cast<float*>(lstats) + 2 * NTYPE * 256), 4, 0, 0);
            }
        }
    };

    f32x16 accW, accA;
#pragma unroll
    for (int e = 0; e < 16; ++e) { accW[e] = 0.f; accA[e] = 0.f; }
    const float c2 = p.scale * 1.4426950408889634f;               // row stats are in log2 units (rowpass_body_bf)
    const bool last_kt = kt * 64 + wk * 32 + 32 > N;              // wave-uniform: only the last key tile holds keys >= N

    // wave-uniform: this wave's 32 x 32 tile lies entirely in the padding (N = 785 leaves 3 of the 4 query sub-tiles of the last
    // query tile and half of the last key tile empty): nothing to score, the wave only keeps the staging and barriers going
    const bool tile_oob = (qt * 128 + wq * 32 >= N) || (kt * 64 + wk * 32 >= N);
    auto score = [&](int offY, int offX, int type, int h, f32x16& acc) {
        if ((EXCEL_DBG(p.dbg) & 1) || tile_oob) return;
        const float2 ml = lstats[((h & 1) * NTYPE + type) * 128 + wq * 32 + r];
        const u16* y16 = tiles + offY + (wk * 32 + r) * 128;
        const u16* x16 = tiles + offX + (wq * 32 + r) * 128;
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int ch = ((s4 * 2 + kh) ^ (r & 15)) * 8, cl = ((8 + s4 * 2 + kh) ^ (r & 15)) * 8;
            const splitx8 yh = *reinterpret_cast<const splitx8*>(y16 + ch), yl = *reinterpret_cast<const splitx8*>(y16 + cl);
            const splitx8 xh = *reinterpret_cast<const splitx8*>(x16 + ch), xl = *reinterpret_cast<const splitx8*>(x16 + cl);
            s = EXCEL_MFMA16(yl, xh, s, 0, 0, 0);
            s = EXCEL_MFMA16(yh, xl, s, 0, 0, 0);
            s = EXCEL_MFMA16(yh, xh, s, 0, 0, 0);
        }
        if (last_kt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kt * 64 + wk * 32 + c32_row(e, lane);
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[e], c2, -ml.x));
                acc[e] += (key < N) ? pr * ml.y : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = fmaf(__builtin_amdgcn_exp2f(fmaf(s[e], c2, -ml.x)), ml.y, acc[e]);
        }
    };

    if (SURGERY) {
        constexpr int XQ = 0, XK = XT, XV = 2 * XT, YQ = 3 * XT, YK = 3 * XT + YT, YV = 3 * XT + 2 * YT;
        issue_y(YQ, 0, 0);
        issue_stats(0);
        issue_x(XQ, 0, 0);
        issue_x(XK, 1, 0);
        issue_y(YK, 1, 0);
        for (int h = 0; h < p.H; ++h) {
            const bool more = h + 1 < p.H && !(EXCEL_DBG(p.dbg) & 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // YQ,stats,XQ(h) landed (XK,YK(h) may be in flight)
            __builtin_amdgcn_s_barrier();
            if (!(EXCEL_DBG(p.dbg) & 2) || h == 0) { issue_x(XV, 2, h); issue_y(YV, 2, h); }
            score(YQ, XQ, 1, h, accA);                                // q.q
            if (!(EXCEL_DBG(p.dbg) & 2) || h == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // XK,YK(h) landed (XV,YV in flight)
            __builtin_amdgcn_s_barrier();
            if (more) { issue_y(YQ, 0, h + 1); issue_stats(h + 1); }
            score(YK, XQ, 0, h, accW);                                // q.k
            __builtin_amdgcn_s_barrier();                             // everyone is done with XQ(h)
            if (more) issue_x(XQ, 0, h + 1);
            score(YK, XK, 2, h, accA);                                // k.k
            if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // XV,YV(h) landed (YQ,stats,XQ(h+1) in flight)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (more) { issue_x(XK, 1, h + 1); issue_y(YK, 1, h + 1); }
            score(YV, XV, 3, h, accA);                                // v.v
        }
    } else {
        // head-mean weights of an nn.MultiheadAttention block: only q.k; (XQ, YK, stats) double buffered
        issue_x(0, 0, 0);
        issue_y(XT, 1, 0);
        issue_stats(0);
        for (int h = 0; h < p.H; ++h) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int cur = (h & 1) * (XT + YT), nxt = (XT + YT) - cur;
            if (h + 1 < p.H) { issue_x(nxt, 0, h + 1); issue_y(nxt + XT, 1, h + 1); issue_stats(h + 1); }
            score(cur + XT, cur, 0, h, accW);
        }
    }
    __syncthreads();
    if (EXCEL_DBG(p.dbg) & 4) { if (accA[0] + accW[0] == 12345.f) p.a_sum[0] = 1.f; return; }

    // transpose each wave's [key][q] tile through LDS (pitch 33) and store rows of q with consecutive keys
    float* tb = reinterpret_cast<float*>(tiles) + wave * (32 * 33);
    const int qbase = qt * 128 + wq * 32, kbase = kt * 64 + wk * 32;
    auto emit = [&](const f32x16& acc, int which) {
#pragma unroll
        for (int e = 0; e < 16; ++e) tb[r * 33 + c32_row(e, lane)] = acc[e];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // unrolled: the 16 read-modify-write round trips on w_aff must be in flight together, not one after the other
        float oldw[16];
        if (which == 1 && p.w_aff && !p.aff_init) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qg = qbase + 2 * i + kh, kg = kbase + r;
                const long long P = N - 1;
                oldw[i] = (qg < N && qg >= 1 && kg >= 1 && kg < N) ? p.w_aff[((long long)b * P + (qg - 1)) * P + (kg - 1)] : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) oldw[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qq = 2 * i + kh;
            const int qg = qbase + qq, kg = kbase + r;
            const float v = tb[qq * 33 + r];
            if (qg >= N) continue;
            if (which == 0) {
                if (kg < p.NP) {
                    float av = v * (1.f / 3.f);
                    // LVC branch (clip_surgery_model.py:140-141): every head's attn[1:,1:] += ex_attn -> head-sum gains H x ex_attn
                    if (p.ex_attn && qg >= 1 && kg >= 1 && kg < N) av += p.ex_scale * p.ex_attn[((long long)b * (N - 1) + (qg - 1)) * (N - 1) + (kg - 1)];
                    if (p.a_sum_split) {
                        split_t* o = reinterpret_cast<split_t*>(p.a_sum) + ((long long)b * N + qg) * 2 * p.NP + split_off(kg, 0);
                        const split_t hi = split_hi(av);
                        o[0] = hi;
                        o[32] = split_hi(av - (float)hi);
                    } else {
                        p.a_sum[((long long)b * N + qg) * p.NP + kg] = av;
                    }
                }
            } else {
                const float pw = v * p.w_scale;
                if (p.attn_out && kg < N) p.attn_out[((long long)b * N + qg) * N + kg] = pw;
                if (p.w_aff && qg >= 1 && kg >= 1 && kg < N) {
                    const long long P = N - 1;
                    p.w_aff[((long long)b * P + (qg - 1)) * P + (kg - 1)] = oldw[i] + pw * p.aff_scale;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    };
    if (SURGERY) emit(accA, 0);
    emit(accW, 1);
}

int excel_launch_attn_rowpass(const float* qkvh, float* out, float* stats, int B, int H, int N, int hd, float scale,
                              int ntypes, hipStream_t st, int split_out, const unsigned short* qkvs, int flash_nq,
                              const unsigned short* vt, int vt_kp) {
    ProfScope prof__(PROF_ATTN_ROWPASS, st);
    EXCEL_CHECK_ARG(hd == HD, "attention: head_dim must be 64 (got %d)", hd);
    EXCEL_CHECK_ARG(ntypes == 1 || ntypes == 4, "attention: ntypes must be 1 or 4");
    RowpassArgs a{qkvh, out, reinterpret_cast<float2*>(stats), B, H, N, scale, split_out, qkvs, flash_nq, vt, vt_kp, 1, 1 << 20};
#ifdef EXCEL_DEV
    { static const int x = getenv("EXCEL_ROWPASS_XCD") ? atoi(getenv("EXCEL_ROWPASS_XCD")) : 1; a.xcd_local = x; }
    { static const int g = getenv("EXCEL_ROWPASS_GRP") ? atoi(getenv("EXCEL_ROWPASS_GRP")) : 0; if (g > 0) a.tail_grp = g; }
#endif
    hipLaunchKernelGGL(attn_rowpass_kernel, dim3(cdiv(N, 128), B * H, ntypes), dim3(256), 0, st, a);
    EXCEL_CHECK_LAUNCH("attn_rowpass");
    return EXCEL_OK;
}

int excel_launch_attn_accum(const float* qkvh, const float* stats, float* a_sum, float* w_aff, float* attn_out, int B, int H,
                            int N, int NP, int hd, float scale, int surgery, float w_scale, float aff_scale, int aff_init,
                            hipStream_t st, const unsigned short* qkvs, int a_sum_split, const float* ex_attn) {
    ProfScope prof__(PROF_ATTN_ACCUM, st);
    EXCEL_CHECK_ARG(hd == HD, "attention: head_dim must be 64 (got %d)", hd);
    EXCEL_CHECK_ARG(!surgery || (a_sum && NP >= N && NP <= cdiv(N, 64) * 64), "attn_accum: bad a_sum/NP");
    AccumArgs a{qkvh, reinterpret_cast<const float2*>(stats), a_sum, w_aff, attn_out, B, H, N, NP, scale, w_scale, aff_scale, aff_init, qkvs, 0, a_sum_split, surgery ? ex_attn : nullptr, (float)H};
    dim3 grid(cdiv(N, 64), cdiv(N, 64), B);
#ifdef EXCEL_DEV
    { static const char* d = getenv("EXCEL_ACCUM_DBG"); if (d) a.dbg = atoi(d); }
#endif
    if (qkvs) {            // bf16x3 mode beyond the strip kernel's reach (attn_strip.hip: N > 1280)
        dim3 g2(cdiv(N, 64), cdiv(N, 128), B);
        if (surgery) hipLaunchKernelGGL((attn_accum_bf_kernel<true>), g2, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((attn_accum_bf_kernel<false>), g2, dim3(512), 0, st, a);
    } else if (surgery)
        hipLaunchKernelGGL((attn_accum_kernel<true, false>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((attn_accum_kernel<false, false>), grid, dim3(256), 0, st, a);
    EXCEL_CHECK_LAUNCH("attn_accum");
    return EXCEL_OK;
}

}  // namespace EXCEL_SPLIT_NS
