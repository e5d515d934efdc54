// Patch-text CAM epilogue (clip/clip.py:288-310 "clip_feature_surgery", else-branch) on top of the
// similarity GEMM  S[b,n,t] = image_features[b,n,:] . text[t,:]  (done by gemm_f32, NT):
//   prob = softmax(2 * S[b,0,:]) ; w = prob / mean(prob)                       (:295-297)
//   sim[n,t] = w[t] S[n,t] - (1/T) sum_t' w[t'] S[n,t']                        (:301-306, GEMM-form identity)
//   attr[n,t] = (sim - min_n sim) / (max_n sim - min_n sim)  over ALL N tokens (:308, quirk Q2)
// and writes the caller's slice [:, 1:, :F] (model/model_excel.py:58) as out[b, n-1, t].
// One 1024-thread workgroup per image: the [N,T] slab is 141 KB (VOC@448) and stays in L2/L1.
#include "common.h"
#include "excel_internal.h"

namespace EXCEL_SPLIT_NS {     // compiled once per 16-bit split type (excel_internal.h, build.py)

#define CAM_TMAX 128

__global__ __launch_bounds__(1024) void cam_epilogue_kernel(float* __restrict__ S, float* __restrict__ out_full,
                                                            float* __restrict__ out_slice, int N, int T, int ldS, int F,
                                                            float temp) {
    __shared__ float w[CAM_TMAX];
    __shared__ float mn[16][CAM_TMAX];
    __shared__ float mx[16][CAM_TMAX];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* Sb = S + (long long)b * N * ldS;

    // --- class-prior weights from the cls token row (wave 0)
    if (wave == 0) {
        float v0 = (lane < T) ? Sb[lane] * temp : -INFINITY;
        float v1 = (lane + 64 < T) ? Sb[lane + 64] * temp : -INFINITY;
        const float m = wave_max(fmaxf(v0, v1));
        float e0 = (lane < T) ? __expf(v0 - m) : 0.f;
        float e1 = (lane + 64 < T) ? __expf(v1 - m) : 0.f;
        const float sum = wave_sum(e0 + e1);
        e0 = e0 / sum;
        e1 = e1 / sum;
        const float mean = wave_sum(e0 + e1) / (float)T;
        if (lane < T) w[lane] = e0 / mean;
        if (lane + 64 < T) w[lane + 64] = e1 / mean;
    }
    __syncthreads();

    // --- sim rows (one wave per token row, lanes over t), in place; per-lane running min/max per column
    const float w0 = (lane < T) ? w[lane] : 0.f;
    const float w1 = (lane + 64 < T) ? w[lane + 64] : 0.f;
    float mn0 = INFINITY, mn1 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY;
    for (int n = wave; n < N; n += 16) {
        float* row = Sb + (long long)n * ldS;
        const float a0 = (lane < T) ? row[lane] * w0 : 0.f;
        const float a1 = (lane + 64 < T) ? row[lane + 64] * w1 : 0.f;
        const float red = wave_sum(a0 + a1) / (float)T;
        const float s0 = a0 - red, s1 = a1 - red;
        if (lane < T) { row[lane] = s0; mn0 = fminf(mn0, s0); mx0 = fmaxf(mx0, s0); }
        if (lane + 64 < T) { row[lane + 64] = s1; mn1 = fminf(mn1, s1); mx1 = fmaxf(mx1, s1); }
    }
    mn[wave][lane] = mn0; mn[wave][lane + 64] = mn1;
    mx[wave][lane] = mx0; mx[wave][lane + 64] = mx1;
    __syncthreads();
    if (tid < CAM_TMAX) {
        float a = mn[0][tid], c = mx[0][tid];
        for (int i = 1; i < 16; ++i) { a = fminf(a, mn[i][tid]); c = fmaxf(c, mx[i][tid]); }
        mn[0][tid] = a;
        mx[0][tid] = c;
    }
    __syncthreads();   // also orders the in-place sim writes before the re-read below (same block)

    const float lo0 = mn[0][lane], lo1 = mn[0][lane + 64];
    const float d0 = mx[0][lane] - lo0, d1 = mx[0][lane + 64] - lo1;
    for (int n = wave; n < N; n += 16) {
        const float* row = Sb + (long long)n * ldS;
        if (lane < T) {
            const float v = (row[lane] - lo0) / d0;
            if (out_full) out_full[((long long)b * N + n) * T + lane] = v;
            if (out_slice && n >= 1 && lane < F) out_slice[((long long)b * (N - 1) + (n - 1)) * F + lane] = v;
        }
        if (lane + 64 < T) {
            const float v = (row[lane + 64] - lo1) / d1;
            if (out_full) out_full[((long long)b * N + n) * T + lane + 64] = v;
            if (out_slice && n >= 1 && lane + 64 < F) out_slice[((long long)b * (N - 1) + (n - 1)) * F + lane + 64] = v;
        }
    }
}

int excel_launch_cam_epilogue(float* S, float* out_full, float* out_slice, int B, int N, int T, int ldS, int F, float temp,
                              hipStream_t st) {
    ProfScope prof__(PROF_CAM_EPILOGUE, st);
    EXCEL_CHECK_ARG(T >= 1 && T <= CAM_TMAX && F <= T && ldS >= T, "cam: need 1 <= F <= T <= %d (T=%d F=%d)", CAM_TMAX, T, F);
    hipLaunchKernelGGL(cam_epilogue_kernel, dim3(B), dim3(1024), 0, st, S, out_full, out_slice, N, T, ldS, F, temp);
    EXCEL_CHECK_LAUNCH("cam_epilogue");
    return EXCEL_OK;
}

// ------------------------------------------------------------------------------------------------ fused patch-text CAM
// The north-star's named kernel: everything between the visual projection and the attribute maps,
//   x_raw [B,N,C] (= ln_post(x) @ proj)  ->  token-axis L2 norm (clip/clip.py:353)  ->  S = f . text^T on the matrix core
//   ->  class-prior weights, redundancy subtraction, min-max over all N tokens (clip/clip.py:288-310)  ->  attr maps.
// Two token-axis reductions bracket the similarity GEMM (column norms before it, per-class min / max after it), so the work is cut
// where they force it and nowhere else - three launches, each over the whole chip:
//   1. colsq_part_kernel: column sums of squares of x_raw over IMAGE-ALIGNED blocks of 64 tokens (a fixed order per image: the maps
//      of an image do not depend on its position in the batch - which is why the partials do not come out of the projection GEMM's
//      epilogue: its row blocks are aligned to the batch, measured and dropped);
//   2. patch_text_sim_kernel, ceil(tiles / 4) workgroups per image x 4 waves, one 32-token tile per wave.  Prologue per workgroup
//      (tiny, L2-resident inputs): inv_norm[c] from the partials, class-prior weights from the cls token on the VALU (fp32 chains,
//      identical bits in every workgroup).  Main: the TRANSPOSED scores S^T[class][token] = text . x'^T (a token is a lane column: the
//      redundancy term, a sum over classes, is an in-lane sum plus one cross-half shuffle), 4 k-steps of operands in flight:
//        BF:  x' = x * inv_norm is split into bf16 hi/lo in registers, text comes pre-split (blocked hi/lo layout of common.h):
//             3 x v_mfma_f32_32x32x16_bf16 per 16-k step (bf16x3, fp32-grade);   !BF: exact fp32 on v_mfma_f32_32x32x2_f32
//      un-normalised similarities go to a [B,N,ldT] scratch (4.5 MB at B = 32), per-workgroup per-class min / max to a partial table;
//   3. patch_text_finish_kernel: min / max over the partials (exact and order-independent), attr = (sim - min) / (max - min), slice.
// The normalised features f are never written unless the caller asks for them (generate_clip_fts' return value).
#define PTC_MAXCT 4            // class tiles of 32: T <= 128
#define PTC_NW 4               // waves (= token tiles) per workgroup
#define PTC_HB 64              // tokens per column-norm block
struct PtcArgs {
    const float* x_raw;            // [B,N,C]
    const float* text;             // [T,C] fp32
    const unsigned short* text_s;  // [T][2C] split bf16 (bf16x3 mode)
    float* sim;                    // [B,N,ldT] scratch: un-normalised similarities
    float* part;                   // [B,G,2,PTC_MAXCT*32] per-workgroup per-class min / max
    float* colsq;                  // [B,cdiv(N,PTC_HB),C] partial column sums of squares
    float* out_full;               // [B,N,T]   (may be null)
    float* out_slice;              // [B,N-1,F] (may be null)
    float* feats;                  // [B,N,C] normalised image_features (may be null)
    int N, C, T, F, ldT, G;
    float temp;
};

// grid (cdiv(N,PTC_HB), B, cdiv(C,256) + 1).  z < cdiv(C,256): part[b][blk][c] = sum of x[b,n,c]^2 over the block's tokens in increasing n
// (one fma chain per column).  Last z slice (bf16x3 mode): the text bank fp32 [T,C] -> split bf16 [T][2C] (blocked hi|lo layout), so
// the split rides in the same launch instead of a serialized 5 us one in front.
__global__ __launch_bounds__(256) void patch_text_prep_kernel(PtcArgs p, unsigned short* text_split_out) {
    const int nzc = gridDim.z - 1;
    if ((int)blockIdx.z == nzc) {
        if (!text_split_out) return;
        const long long i = (((long long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x) * 4;
        const long long stride = (long long)gridDim.x * gridDim.y * 256 * 4;
        for (long long e = i; e < (long long)p.T * p.C; e += stride) {
            const long long row = e / p.C;
            const int k = (int)(e - row * p.C);
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.text + e);
            split_t hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { hi[j] = split_hi(v[j]); lo[j] = split_hi(v[j] - (float)hi[j]); }
            split_t* o = reinterpret_cast<split_t*>(text_split_out) + row * 2 * p.C + split_off(k, 0);
            *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(hi);
            *reinterpret_cast<uint2*>(o + 32) = *reinterpret_cast<const uint2*>(lo);
        }
        return;
    }
    const int blk = blockIdx.x, b = blockIdx.y, c = blockIdx.z * 256 + threadIdx.x;
    if (c >= p.C) return;
    const int n0 = blk * PTC_HB, n1 = min(n0 + PTC_HB, p.N);
    const float* x = p.x_raw + ((long long)b * p.N + n0) * p.C + c;
    float s = 0.f;
#pragma unroll 8
    for (int n = n0; n < n1; ++n, x += p.C) s = fmaf(*x, *x, s);
    p.colsq[((long long)b * gridDim.x + blk) * p.C + c] = s;
}

__device__ __forceinline__ float half_min(float v) {          // over the 32 lanes of a wave half
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// TLDS: the split text bank (T <= PTC_TLDS_ROWS rows of 2C bf16) is staged in LDS once per workgroup - the four waves multiply against
// the same rows, and fragment loads of 16 bytes per lane from 32 different rows are address-bound in the vector memory path (measured:
// 2/3 of the kernel's 41 us with text and x both read as per-lane row pieces).  x_raw tiles are read COALESCED (8 lanes = one 128-byte
// row piece) and turned into MFMA operand order through a wave-private LDS tile.
#define PTC_TLDS_ROWS 48
#define PTC_TLDS_C 512
#define PTC_XP 36              // LDS pitch (floats) of the 32 x 32 x-tile: conflict-free 16-byte operand reads
#define PTC_PFB 4              // 32-column blocks of x in flight per wave
template <bool BF, int CT, bool TLDS>
__global__ __launch_bounds__(PTC_NW * 64) void patch_text_sim_kernel(PtcArgs p) {
    constexpr int NW = PTC_NW, NT = NW * 64;
    constexpr int TPITCH = 2 * PTC_TLDS_C + 8;                 // bf16 per LDS text row (C <= PTC_TLDS_C), +16 B against bank conflicts
    __shared__ __attribute__((aligned(16))) float inv[1024];   // 1 / ||x[:, c]||_2 over the tokens
    __shared__ __attribute__((aligned(16))) float f0[1024];    // normalised cls-token features
    __shared__ float w[PTC_MAXCT * 32];            // cls logits, then the class-prior weights (0 for padded classes)
    __shared__ float red_mn[NW][PTC_MAXCT * 32], red_mx[NW][PTC_MAXCT * 32];
    __shared__ __attribute__((aligned(16))) float xs[NW][32 * PTC_XP];
    constexpr int RAW_TEXT = TLDS ? PTC_TLDS_ROWS * TPITCH * 2 : 0, RAW_EP = NW * 32 * (CT * 32 + 1) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char raw[RAW_TEXT > RAW_EP ? RAW_TEXT : RAW_EP];   // text bank, then the epilogue tiles
    unsigned short* tsm = reinterpret_cast<unsigned short*>(raw);
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kh = lane >> 5;
    const int N = p.N, C = p.C, T = p.T;
    const float* X = p.x_raw + (long long)b * N * C;
    const int ntile = (N + 31) / 32;
    const int tile = g * NW + wave;
    const bool live = tile < ntile;
    // coalesced x loads: lane -> 16 bytes of row (lane >> 3) + 8 j at column (lane & 7) * 4 of a 32-column block
    const int lr = lane >> 3, lc = (lane & 7) * 4;
    const float* xq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xq[j] = X + (long long)min(tile * 32 + lr + 8 * j, N - 1) * C + lc;    // clamped rows: masked at the end
    const int nkb = C / 32;
    f32x4 xv[PTC_PFB][4];
    // the first blocks of the wave's tile are issued BEFORE the prologue (they do not depend on it): the prologue's trips to L2 and the
    // first trip to x_raw overlap
    if (live) {
#pragma unroll
        for (int s = 0; s < PTC_PFB; ++s)
            if (s < nkb) {
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[s][j] = *reinterpret_cast<const f32x4*>(xq[j] + s * 32);
            }
    }
    // ---- prologue: text bank -> LDS and the token-axis norms (partials in block order).  Every load of the prologue is issued before the
    // first wait (the workgroup is one chain of dependent trips to L2: each avoided trip is ~0.7 us of a ~20 us kernel)
    {
        // text bank -> LDS by LDS-DMA (global_load_lds, 16 B per lane, lane-linear destination): one wave-instruction = 1 KB of a split
        // row, no registers and no wait until the single vmcnt(0) below (a register-staged copy is sunk by the compiler into one
        // dependent round trip per chunk: 24 trips)
        if (TLDS && BF) {
            const int per_row = (2 * C * 2) / 1024;             // 1-KB pieces per split row (C % 256 == 0 on this path)
            for (int i = wave; i < T * per_row; i += NW) {
                const int row = i / per_row, pc = i - row * per_row;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.text_s + (long long)row * 2 * C + pc * 512 + lane * 8),
                                                 (__attribute__((address_space(3))) void*)(tsm + row * TPITCH + pc * 512), 16, 0, 0);
            }
        }
        const int nblk = (N + PTC_HB - 1) / PTC_HB;
        const float* q = p.colsq + (long long)b * nblk * C;
        for (int c = tid; c < C; c += NT) {
            float s = 0.f;
            for (int k0 = 0; k0 < nblk; k0 += 8) {             // 8 partials in flight; added in block order (s + 0 is exact for the tail)
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = q[(long long)min(k0 + k, nblk - 1) * C + c];
#pragma unroll
                for (int k = 0; k < 8; ++k) s += (k0 + k < nblk) ? v[k] : 0.f;
            }
            inv[c] = 1.f / sqrtf(s);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the LDS-DMA pieces (and the x prefetch) have landed
        for (int t = tid; t < PTC_MAXCT * 32; t += NT) w[t] = 0.f;
    }
    __syncthreads();
    for (int c = tid; c < C; c += NT) f0[c] = X[c] * inv[c];    // normalised cls-token features (used after the main loop)

    float* simb = p.sim + (long long)b * N * p.ldT;
    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
    if (live) {
        const int ntok = min(tile * 32 + r, N - 1);           // token of this lane column in operand order
        float* frow = (p.feats && tile * 32 + r < N) ? p.feats + ((long long)b * N + ntok) * C : nullptr;
        float* xt = xs[wave];
        // text operand rows of this lane (padded class rows: clamped, zeroed below)
        const unsigned short* trow[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int cls = min(ct * 32 + r, T - 1);
            trow[ct] = TLDS ? &tsm[cls * TPITCH] : p.text_s + (long long)cls * 2 * C;
        }
        for (int kb0 = 0; kb0 < nkb; kb0 += PTC_PFB) {
#pragma unroll
            for (int s = 0; s < PTC_PFB; ++s) {
                const int kb = kb0 + s;
                if (kb < nkb) {
                // coalesced registers -> LDS tile (wave-private) -> operand order
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&xt[(lr + 8 * j) * PTC_XP + lc]) = xv[s][j];
                {   // refill the slot just consumed - unconditionally (the tail re-reads the last block, an L1 hit): a load under a
                    // branch makes the wait-count pass assume it may be missing and turns every counted vmcnt into a full drain
                    const int kn = min(kb + PTC_PFB, nkb - 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[s][j] = *reinterpret_cast<const f32x4*>(xq[j] + kn * 32);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int cb = h2 * 16 + 8 * kh, c0 = kb * 32 + cb;       // this lane's 8 columns of the 16-k step
                    const f32x4 ia = *reinterpret_cast<const f32x4*>(inv + c0), ib = *reinterpret_cast<const f32x4*>(inv + c0 + 4);
                    const f32x4 fa = *reinterpret_cast<const f32x4*>(&xt[r * PTC_XP + cb]) * ia;
                    const f32x4 fb = *reinterpret_cast<const f32x4*>(&xt[r * PTC_XP + cb + 4]) * ib;
                    if (frow) { *reinterpret_cast<f32x4*>(frow + c0) = fa; *reinterpret_cast<f32x4*>(frow + c0 + 4) = fb; }
                    if (BF) {
                        splitx8 xh, xl;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float v = (j < 4) ? fa[j] : fb[j - 4];
                            xh[j] = split_hi(v);
                            xl[j] = split_hi(v - (float)xh[j]);
                        }
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) {
                            const unsigned short* tp = trow[ct] + split_off(c0, 0);
                            splitx8 h = *reinterpret_cast<const splitx8*>(tp), l = *reinterpret_cast<const splitx8*>(tp + 32);
                            if (ct * 32 + r >= T) { h = splitx8{0, 0, 0, 0, 0, 0, 0, 0}; l = h; }
                            acc[ct] = EXCEL_MFMA16(l, xh, acc[ct], 0, 0, 0);
                            acc[ct] = EXCEL_MFMA16(h, xl, acc[ct], 0, 0, 0);
                            acc[ct] = EXCEL_MFMA16(h, xh, acc[ct], 0, 0, 0);
                        }
                    } else {
                        // exact fp32: v_mfma_f32_32x32x2_f32 takes k = {kh}: feed the lane's 8 columns as 8 k-pairs (kh selects the column
                        // within a pair consistently for both operands: any consistent k order gives the same fma chain per output)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) {
                            const int cls = ct * 32 + r;
                            f32x4 ta = {0.f, 0.f, 0.f, 0.f}, tb = ta;
                            if (cls < T) {
                                ta = *reinterpret_cast<const f32x4*>(p.text + (long long)cls * C + c0);
                                tb = *reinterpret_cast<const f32x4*>(p.text + (long long)cls * C + c0 + 4);
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[e], fa[e], acc[ct], 0, 0, 0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(tb[e], fb[e], acc[ct], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();              // the tile is re-written by the next block
                }
            }
        }
    }
    // ---- class-prior weights  softmax(temp * S[0,:]) / mean  (clip.py:295-297), AFTER the matrix loop (it needs them only in the
    // epilogue).  S[0,t] = sum_c f[0,c] text[t,c]: thread = (class, quarter of the 16-byte pieces of a row), fixed fp32 fma chains
    // (identical bits in every workgroup and for every batch composition).
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += NT / 4) {
        const int t = t0 + (tid >> 2), q = tid & 3;
        float s = 0.f;
        if (t < T) {
            const float* tr = p.text + (long long)t * C;
#pragma unroll 8
            for (int c = q * 4; c < C; c += 16) {
                const f32x4 tv = *reinterpret_cast<const f32x4*>(tr + c), fv = *reinterpret_cast<const f32x4*>(f0 + c);
                s = fmaf(fv[0], tv[0], s); s = fmaf(fv[1], tv[1], s); s = fmaf(fv[2], tv[2], s); s = fmaf(fv[3], tv[3], s);
            }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (t < T && q == 0) w[t] = s;
    }
    __syncthreads();
    if (wave == 0) {
        float v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i] = (lane + 64 * i < T) ? w[lane + 64 * i] * p.temp : -INFINITY;
        const float m = wave_max(fmaxf(v[0], v[1]));
        float e0 = (lane < T) ? __expf(v[0] - m) : 0.f, e1 = (lane + 64 < T) ? __expf(v[1] - m) : 0.f;
        const float sum = wave_sum(e0 + e1);
        e0 = e0 / sum;
        e1 = e1 / sum;
        const float mean = wave_sum(e0 + e1) / (float)T;
        __builtin_amdgcn_wave_barrier();
        w[lane] = (lane < T) ? e0 / mean : 0.f;
        w[lane + 64] = (lane + 64 < T) ? e1 / mean : 0.f;
    }
    __syncthreads();
    // sim[n,t] = w[t] S[n,t] - (1/T) sum_t' w[t'] S[n,t']      (clip.py:301-306 in GEMM form); per-class min / max of this tile
    float part = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc[ct][e] *= w[ct * 32 + c32_row(e, lane)];
            part += acc[ct][e];
        }
    part += __shfl_xor(part, 32, 64);
    const float red = part / (float)T;
    // the tile goes through LDS as [token][class]: lanes then run over the CLASSES - a token row of T similarities is one coalesced
    // store, and a class's min / max over the tile's tokens is an in-lane loop (320 cross-lane shuffles per wave otherwise).  The text
    // bank's LDS is reused for the tiles: every wave passed the barriers of the class-prior block after its matrix loop
    constexpr int EP = CT * 32 + 1;
    float* et = reinterpret_cast<float*>(raw) + wave * (32 * EP);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) et[r * EP + ct * 32 + c32_row(e, lane)] = acc[ct][e] - red;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int nvalid = live ? min(32, N - tile * 32) : 0;    // tokens of this tile inside the image
#pragma unroll
    for (int cc = 0; cc < (CT + 1) / 2; ++cc) {
        const int cls = cc * 64 + lane;
        float a = INFINITY, c = -INFINITY;
        if (cls < T) {
            float* srow = simb + (long long)tile * 32 * p.ldT + cls;
#pragma unroll 8
            for (int k = 0; k < nvalid; ++k) {
                const float v = et[k * EP + cls];
                srow[(long long)k * p.ldT] = v;
                a = fminf(a, v);
                c = fmaxf(c, v);
            }
        }
        if (cls < CT * 32) { red_mn[wave][cls] = a; red_mx[wave][cls] = c; }
    }
    __syncthreads();
    if (tid < CT * 32) {
        float a = red_mn[0][tid], c = red_mx[0][tid];
#pragma unroll
        for (int i = 1; i < NW; ++i) { a = fminf(a, red_mn[i][tid]); c = fmaxf(c, red_mx[i][tid]); }
        float* o = p.part + ((long long)b * p.G + g) * 2 * (PTC_MAXCT * 32);
        o[tid] = a;
        o[PTC_MAXCT * 32 + tid] = c;
    }
}

// attr = (sim - min) / (max - min) over ALL tokens (clip.py:308; NaN when max == min, like the reference).  grid (cdiv(N, 64), B)
__global__ __launch_bounds__(256) void patch_text_finish_kernel(PtcArgs p) {
    __shared__ float lo[PTC_MAXCT * 32], span[PTC_MAXCT * 32];
    const int b = blockIdx.y, tid = threadIdx.x, N = p.N, T = p.T;
    if (tid < T) {
        const float* q = p.part + (long long)b * p.G * 2 * (PTC_MAXCT * 32);
        float a = INFINITY, c = -INFINITY;
#pragma unroll 8
        for (int g = 0; g < p.G; ++g) { a = fminf(a, q[g * 2 * (PTC_MAXCT * 32) + tid]); c = fmaxf(c, q[(g * 2 + 1) * (PTC_MAXCT * 32) + tid]); }
        lo[tid] = a;
        span[tid] = c - a;
    }
    __syncthreads();
    const float* simb = p.sim + (long long)b * N * p.ldT;
    const int n0 = blockIdx.x * 64, n1 = min(n0 + 64, N);
#pragma unroll 4
    for (int i = tid; i < (n1 - n0) * T; i += 256) {
        const int n = n0 + i / T, t = i % T;
        const float v = (simb[(long long)n * p.ldT + t] - lo[t]) / span[t];
        if (p.out_full) p.out_full[((long long)b * N + n) * T + t] = v;
        if (p.out_slice && n >= 1 && t < p.F) p.out_slice[((long long)b * (N - 1) + (n - 1)) * p.F + t] = v;
    }
}

static inline int ptc_groups(int N) { return cdiv(cdiv(N, 32), PTC_NW); }
size_t excel_patch_text_cam_ws_floats(int B, int N, int C, int T, int which) {
    if (which == 0) return (size_t)B * N * ((T + 3) / 4 * 4);
    if (which == 1) return (size_t)B * ptc_groups(N) * 2 * (PTC_MAXCT * 32);
    return (size_t)B * cdiv(N, PTC_HB) * C;
}

// text_split (bf16x3 mode): workspace of T*C floats the prep kernel fills with the split text bank
int excel_launch_patch_text_cam(const float* x_raw, const float* text, unsigned short* text_split_out, float* sim_ws, float* part_ws,
                                float* colsq_ws, float* out_full, float* out_slice, float* feats, int B, int N, int C, int T, int F, int ldT,
                                float temp, int bf, hipStream_t st) {
    ProfScope prof__(PROF_CAM_FUSED, st, 2.0 * B * (double)N * C * T);
    EXCEL_CHECK_ARG(T >= 1 && T <= PTC_MAXCT * 32 && F <= T && ldT >= T, "patch_text_cam: need 1 <= F <= T <= %d (T=%d F=%d)", PTC_MAXCT * 32, T, F);
    EXCEL_CHECK_ARG(C <= 1024 && (C % 32) == 0, "patch_text_cam: C must be a multiple of 32, <= 1024 (C=%d)", C);
    EXCEL_CHECK_ARG(!bf || text_split_out, "patch_text_cam: bf16x3 mode needs the split-text workspace");
    const unsigned short* text_split = text_split_out;
    // the kernels read x_raw / text / the split text and write feats with 16-byte vectors
    EXCEL_CHECK_ARG((((uintptr_t)x_raw | (uintptr_t)text | (uintptr_t)text_split | (uintptr_t)feats | (uintptr_t)sim_ws) & 15) == 0,
                    "patch_text_cam: x_raw, text, image_features and the workspace must be 16-byte aligned");
    const int G = ptc_groups(N);
    PtcArgs a{x_raw, text, text_split, sim_ws, part_ws, colsq_ws, out_full, out_slice, feats, N, C, T, F, ldT, G, temp};
    hipLaunchKernelGGL(patch_text_prep_kernel, dim3(cdiv(N, PTC_HB), B, cdiv(C, 256) + 1), dim3(256), 0, st, a, bf ? text_split_out : nullptr);
    const int ct = cdiv(T, 32);
#define PTC_LAUNCH(BFV, CTV, TL) hipLaunchKernelGGL((patch_text_sim_kernel<BFV, CTV, TL>), dim3(G, B), dim3(PTC_NW * 64), 0, st, a)
    if (bf) {
        if (T <= PTC_TLDS_ROWS && C <= PTC_TLDS_C && (C % 256) == 0) { if (ct <= 1) PTC_LAUNCH(true, 1, true); else PTC_LAUNCH(true, 2, true); }
        else { if (ct <= 2) PTC_LAUNCH(true, 2, false); else PTC_LAUNCH(true, 4, false); }
    } else { if (ct <= 1) PTC_LAUNCH(false, 1, false); else if (ct == 2) PTC_LAUNCH(false, 2, false); else PTC_LAUNCH(false, 4, false); }
#undef PTC_LAUNCH
    hipLaunchKernelGGL(patch_text_finish_kernel, dim3(cdiv(N, 64), B), dim3(256), 0, st, a);
    EXCEL_CHECK_LAUNCH("patch_text_cam");
    return EXCEL_OK;
}

}  // namespace EXCEL_SPLIT_NS
