// Patch-text CAM epilogue (clip/clip.py:288-310 "clip_feature_surgery", else-branch) on top of the
// similarity GEMM  S[b,n,t] = image_features[b,n,:] . text[t,:]  (done by gemm_f32, NT):
//   prob = softmax(2 * S[b,0,:]) ; w = prob / mean(prob)                       (:295-297)
//   sim[n,t] = w[t] S[n,t] - (1/T) sum_t' w[t'] S[n,t']                        (:301-306, GEMM-form identity)
//   attr[n,t] = (sim - min_n sim) / (max_n sim - min_n sim)  over ALL N tokens (:308, quirk Q2)
// and writes the caller's slice [:, 1:, :F] (model/model_excel.py:58) as out[b, n-1, t].
// One 1024-thread workgroup per image: the [N,T] slab is 141 KB (VOC@448) and stays in L2/L1.
#include "common.h"
#include "excel_internal.h"

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
// The north-star's named kernel: everything between the visual projection and the attribute maps in ONE launch,
//   x_raw [B,N,C] (= ln_post(x) @ proj)  ->  token-axis L2 norm (clip/clip.py:353)  ->  S = f . text^T on the matrix core
//   ->  class-prior weights, redundancy subtraction, min-max over all N tokens (clip/clip.py:288-310)  ->  attr maps.
// One workgroup per image (the reductions over the token axis - column norms and per-class min/max - stay inside it); the
// 8 waves take 32-token tiles.  Per tile a wave computes the TRANSPOSED scores S^T[class][token] = text . x'^T, so a token is a
// lane column: the redundancy term (a sum over classes) is an in-lane sum plus one cross-half shuffle.
//   BF:  x' = x * inv_norm is split into bf16 hi/lo in registers, text comes pre-split (blocked hi/lo layout of common.h):
//        3 x v_mfma_f32_32x32x16_bf16 per 16-k step (bf16x3, fp32-grade)
//   !BF: exact fp32 on v_mfma_f32_32x32x2_f32
// The normalised features f are never written unless the caller asks for them (generate_clip_fts' return value).
#define PTC_MAXCT 4            // class tiles of 32: T <= 128
struct PtcArgs {
    const float* x_raw;            // [B,N,C]
    const float* text;             // [T,C] fp32 (exact mode)
    const unsigned short* text_s;  // [T][2C] split bf16 (bf16x3 mode)
    float* sim;                    // [B,N,ldT] scratch: un-normalised similarities
    float* out_full;               // [B,N,T]   (may be null)
    float* out_slice;              // [B,N-1,F] (may be null)
    float* feats;                  // [B,N,C] normalised image_features (may be null)
    int N, C, T, F, ldT;
    float temp;
};

// NT threads per workgroup: 1024 where the register budget of 16 waves allows it (<= 2 class tiles), 512 otherwise.  The workgroup is
// latency-bound (one CU streams its image's [N, C] slab twice), so the wave count is its throughput.
template <bool BF, int CT, int NT>
__global__ __launch_bounds__(NT) void patch_text_cam_kernel(PtcArgs p) {
    constexpr int NW = NT / 64;
    __shared__ float inv[1024];                    // 1 / ||x[:, c]||_2 over the tokens
    __shared__ float w[PTC_MAXCT * 32];            // class-prior weights (0 for padded classes)
    __shared__ float red_mn[NW][PTC_MAXCT * 32], red_mx[NW][PTC_MAXCT * 32];
    __shared__ float part[16 * 512];               // [residue n % 16][column] partial sums of squares (columns in chunks of <= 512)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kh = lane >> 5;
    const int N = p.N, C = p.C, T = p.T;
    const float* X = p.x_raw + (long long)b * N * C;

    // ---- column norms over the token axis, fixed summation order: 16 partial sums per column (rows n = j mod 16, increasing n),
    // combined pairwise.  The 16 residues of a column are spread over G = NT / columns threads (more row loads in flight: the loop
    // is latency-bound); the partials meet in LDS and the pairwise tree is the same whatever G is.
    for (int c0 = 0; c0 < C; c0 += 512) {
        const int cw = min(512, C - c0);
        int G = 1;
        while (G < 16 && 2 * G * cw <= NT) G *= 2;
        const int per = 16 / G;                                   // residues per thread
        if (tid < G * cw) {
            const int g = tid / cw, c = c0 + tid - g * cw;
            float s16[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) s16[j] = 0.f;
            for (int n0 = 0; n0 < N; n0 += 16) {
                float v[16];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const int n = n0 + g * per + jj;
                    v[jj] = (jj < per && n < N) ? X[(long long)n * C + c] : 0.f;
                }
#pragma unroll
                for (int jj = 0; jj < 16; ++jj)
                    if (jj < per && n0 + g * per + jj < N) s16[jj] = fmaf(v[jj], v[jj], s16[jj]);
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj)
                if (jj < per) part[(g * per + jj) * 512 + (c - c0)] = s16[jj];
        }
        __syncthreads();
        if (tid < cw) {
            float s16[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) s16[j] = part[j * 512 + tid];
#pragma unroll
            for (int st = 8; st > 0; st >>= 1)
#pragma unroll
                for (int j = 0; j < st; ++j) s16[j] += s16[j + st];
            inv[c0 + tid] = 1.f / sqrtf(s16[0]);
        }
        __syncthreads();
    }
    __syncthreads();
    if (p.feats) {
        float* Fo = p.feats + (long long)b * N * C;
        for (long long i = (long long)tid * 4; i < (long long)N * C; i += NT * 4) {
            const int c = (int)(i % C);
            f32x4 v = *reinterpret_cast<const f32x4*>(X + i);
            v[0] *= inv[c]; v[1] *= inv[c + 1]; v[2] *= inv[c + 2]; v[3] *= inv[c + 3];
            *reinterpret_cast<f32x4*>(Fo + i) = v;
        }
    }

    float* simb = p.sim + (long long)b * N * p.ldT;
    const int ntile = (N + 31) / 32;
    bool first = true;
    for (int tile = wave; tile < ntile || first; tile += NW) {
        const bool live = tile < ntile;                       // every wave takes part in the barrier of its first round
        const int n = min(tile * 32 + r, N - 1);              // token of this lane column (clamped; masked at the end)
        f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
        if (live) {
            const float* xrow = X + (long long)n * C;
            if (BF) {
                // register double buffer: the loads of k-step k0+16 are in flight while k0 is split and multiplied
                f32x4 xa, xb;
                bf16x8 th[CT], tl[CT];
                auto fetch = [&](int k0, f32x4& a, f32x4& bq, bf16x8 (&h)[CT], bf16x8 (&l)[CT]) {
                    const int c0 = k0 + 8 * kh;
                    a = *reinterpret_cast<const f32x4*>(xrow + c0);
                    bq = *reinterpret_cast<const f32x4*>(xrow + c0 + 4);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int cls = min(ct * 32 + r, T - 1);                      // padded class rows: clamped load, zeroed below
                        const unsigned short* tp = p.text_s + (long long)cls * 2 * C + split_off(c0, 0);
                        h[ct] = *reinterpret_cast<const bf16x8*>(tp);
                        l[ct] = *reinterpret_cast<const bf16x8*>(tp + 32);
                    }
                };
                fetch(0, xa, xb, th, tl);
                for (int k0 = 0; k0 < C; k0 += 16) {
                    f32x4 nxa = xa, nxb = xb;
                    bf16x8 nth[CT], ntl[CT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) { nth[ct] = th[ct]; ntl[ct] = tl[ct]; }
                    if (k0 + 16 < C) fetch(k0 + 16, nxa, nxb, nth, ntl);
                    const int c0 = k0 + 8 * kh;
                    const f32x4 ia = *reinterpret_cast<const f32x4*>(inv + c0), ib = *reinterpret_cast<const f32x4*>(inv + c0 + 4);
                    bf16x8 xh, xl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = (j < 4) ? xa[j] * ia[j] : xb[j - 4] * ib[j - 4];
                        xh[j] = (__bf16)v;
                        xl[j] = (__bf16)(v - (float)xh[j]);
                    }
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        bf16x8 h = th[ct], l = tl[ct];
                        if (ct * 32 + r >= T) { h = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; l = h; }
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l, xh, acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, xl, acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, xh, acc[ct], 0, 0, 0);
                    }
                    xa = nxa; xb = nxb;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) { th[ct] = nth[ct]; tl[ct] = ntl[ct]; }
                }
            } else {
                for (int k0 = 0; k0 < C; k0 += 8) {
                    const int c0 = k0 + 4 * kh;
                    f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + c0);
                    const f32x4 iv = *reinterpret_cast<const f32x4*>(inv + c0);
                    xv *= iv;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int cls = ct * 32 + r;
                        f32x4 tv = {0.f, 0.f, 0.f, 0.f};
                        if (cls < T) tv = *reinterpret_cast<const f32x4*>(p.text + (long long)cls * C + c0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(tv[e], xv[e], acc[ct], 0, 0, 0);
                    }
                }
            }
        }
        if (first) {
            // class-prior weights from the cls token (token 0 = lane columns 0 / 32 of wave 0's first tile):  softmax(temp * S[0,:]) / mean
            if (wave == 0) {
                if (r == 0) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int e = 0; e < 16; ++e) w[ct * 32 + c32_row(e, lane)] = acc[ct][e];      // S[0, class]
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
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
                if (CT > 2) w[lane + 64] = (lane + 64 < T) ? e1 / mean : 0.f;
            }
            __syncthreads();
            first = false;
        }
        if (!live) break;
        // sim[n,t] = w[t] S[n,t] - (1/T) sum_t' w[t'] S[n,t']      (clip.py:301-306 in GEMM form)
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
        const bool tok_ok = tile * 32 + r < N;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cls = ct * 32 + c32_row(e, lane);
                const float s = acc[ct][e] - red;
                if (tok_ok && cls < T) simb[(long long)(tile * 32 + r) * p.ldT + cls] = s;
            }
    }
    __syncthreads();       // orders this workgroup's sim writes before the re-reads below
    // ---- per-class min / max over all tokens (one wave per token row, lanes over the classes; the [N,T] slab is L2/L1 resident)
    {
        float mn0 = INFINITY, mn1 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY;
        for (int n = wave; n < N; n += NW) {
            const float* row = simb + (long long)n * p.ldT;
            if (lane < T) { const float v = row[lane]; mn0 = fminf(mn0, v); mx0 = fmaxf(mx0, v); }
            if (lane + 64 < T) { const float v = row[lane + 64]; mn1 = fminf(mn1, v); mx1 = fmaxf(mx1, v); }
        }
        red_mn[wave][lane] = mn0; red_mx[wave][lane] = mx0;
        red_mn[wave][lane + 64] = mn1; red_mx[wave][lane + 64] = mx1;
    }
    __syncthreads();
    if (tid < PTC_MAXCT * 32) {
        float a = red_mn[0][tid], c = red_mx[0][tid];
        for (int i = 1; i < NW; ++i) { a = fminf(a, red_mn[i][tid]); c = fmaxf(c, red_mx[i][tid]); }
        red_mn[0][tid] = a;
        red_mx[0][tid] = c - a;
    }
    __syncthreads();
    // ---- attr = (sim - min) / (max - min)   (clip.py:308; NaN when max == min, like the reference)
    for (long long i = tid; i < (long long)N * T; i += NT) {
        const int n = (int)(i / T), t = (int)(i - (long long)n * T);
        const float v = (simb[(long long)n * p.ldT + t] - red_mn[0][t]) / red_mx[0][t];
        if (p.out_full) p.out_full[((long long)b * N + n) * T + t] = v;
        if (p.out_slice && n >= 1 && t < p.F) p.out_slice[((long long)b * (N - 1) + (n - 1)) * p.F + t] = v;
    }
}

int excel_launch_patch_text_cam(const float* x_raw, const float* text, const unsigned short* text_split, float* sim_ws, float* out_full,
                                float* out_slice, float* feats, int B, int N, int C, int T, int F, int ldT, float temp, int bf, hipStream_t st) {
    ProfScope prof__(PROF_CAM_FUSED, st, 2.0 * B * (double)N * C * T);
    EXCEL_CHECK_ARG(T >= 1 && T <= PTC_MAXCT * 32 && F <= T && ldT >= T, "patch_text_cam: need 1 <= F <= T <= %d (T=%d F=%d)", PTC_MAXCT * 32, T, F);
    EXCEL_CHECK_ARG(C <= 1024 && (C % 32) == 0, "patch_text_cam: C must be a multiple of 32, <= 1024 (C=%d)", C);
    EXCEL_CHECK_ARG(!bf || text_split, "patch_text_cam: bf16x3 mode needs the split text");
    PtcArgs a{x_raw, text, text_split, sim_ws, out_full, out_slice, feats, N, C, T, F, ldT, temp};
    const int ct = cdiv(T, 32);
#define PTC_LAUNCH(BFV, CTV, NTV) hipLaunchKernelGGL((patch_text_cam_kernel<BFV, CTV, NTV>), dim3(B), dim3(NTV), 0, st, a)
    if (bf) { if (ct <= 1) PTC_LAUNCH(true, 1, 1024); else if (ct == 2) PTC_LAUNCH(true, 2, 1024); else PTC_LAUNCH(true, 4, 512); }
    else { if (ct <= 1) PTC_LAUNCH(false, 1, 1024); else if (ct == 2) PTC_LAUNCH(false, 2, 1024); else PTC_LAUNCH(false, 4, 512); }
#undef PTC_LAUNCH
    EXCEL_CHECK_LAUNCH("patch_text_cam");
    return EXCEL_OK;
}
