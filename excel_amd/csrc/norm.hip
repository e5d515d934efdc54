// Row/column normalisation kernels of the ViT path (all HBM-bound, one pass over the data).
//   layernorm_rows      : LayerNorm (fp32, eps 1e-5; clip/clip_surgery_model.py:271-277), one wave per token row
//   assemble_ln_pre     : [cls | patch-embed GEMM out] + positional embedding -> ln_pre (:424-438), fused
//   token_axis_normalize: image_features / ||.||_2 over the TOKEN axis (clip/clip.py:353; quirk Q1)
//   im2col              : patch gather for the stride-16 conv1 (:421)
#include "common.h"
#include "excel_internal.h"

namespace EXCEL_SPLIT_NS {     // compiled once per 16-bit split type (excel_internal.h, build.py)

__device__ __forceinline__ void ln_row(const float* __restrict__ src, const float* __restrict__ add,
                                       const float* __restrict__ w, const float* __restrict__ b,
                                       float* __restrict__ dst, int D, float eps, int lane, bool split = false) {
    // pass 1: mean (the row is 3 KB: passes 2 and 3 hit L1)
    float s = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
        if (add) v += *reinterpret_cast<const f32x4*>(add + c);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = wave_sum(s) / (float)D;
    // pass 2: biased variance of (x - mean)
    float q = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
        if (add) v += *reinterpret_cast<const f32x4*>(add + c);
        v -= mean;
        q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
        if (add) v += *reinterpret_cast<const f32x4*>(add + c);
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
        v = (v - mean) * rstd * ww + bb;
        if (split) {   // bf16 hi/lo planes [2][D] in the same D*4 bytes (operand format of the bf16x3 GEMM)
            split_t hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { hi[j] = split_hi(v[j]); lo[j] = split_hi(v[j] - (float)hi[j]); }
            split_t* o = reinterpret_cast<split_t*>(dst);
            *reinterpret_cast<uint2*>(o + split_off(c, 0)) = *reinterpret_cast<const uint2*>(hi);
            *reinterpret_cast<uint2*>(o + split_off(c, 1)) = *reinterpret_cast<const uint2*>(lo);
        } else {
            *reinterpret_cast<f32x4*>(dst + c) = v;
        }
    }
}

// The same row with D = 256 * NCH held in registers (one global read of the row instead of three passes through L1; same operations in
// the same order -> same bits).  D = 768: NCH = 3.
template <int NCH>
__device__ __forceinline__ void ln_row_reg(const float* __restrict__ src, const float* __restrict__ add, const float* __restrict__ w,
                                           const float* __restrict__ b, float* __restrict__ dst, float eps, int lane, bool split) {
    constexpr int D = 256 * NCH;
    f32x4 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(src + lane * 4 + 256 * i);
        if (add) v[i] += *reinterpret_cast<const f32x4*>(add + lane * 4 + 256 * i);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        v[i] -= mean;
        q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane * 4 + 256 * i;
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
        const f32x4 o4 = v[i] * rstd * ww + bb;
        if (split) {
            split_t hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { hi[j] = split_hi(o4[j]); lo[j] = split_hi(o4[j] - (float)hi[j]); }
            split_t* o = reinterpret_cast<split_t*>(dst);
            *reinterpret_cast<uint2*>(o + split_off(c, 0)) = *reinterpret_cast<const uint2*>(hi);
            *reinterpret_cast<uint2*>(o + split_off(c, 1)) = *reinterpret_cast<const uint2*>(lo);
        } else {
            *reinterpret_cast<f32x4*>(dst + c) = o4;
        }
    }
}

// y[row] = LN(x[row]);  if cls_src != null, rows with (row % tokN == 0) read from cls_src instead
// (the x[0] = x_ori[0] swap of clip_surgery_model.py:442, fused into ln_post).
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ cls_src,
                                                             int tokN, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ y,
                                                             int rows, int D, float eps, int split_out, long long in_stride) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* src = x + (long long)row * in_stride;      // in_stride != D: gather every tokN-th row (cls tokens), compact output
    if (cls_src && (row % tokN) == 0) src = cls_src + (long long)row * in_stride;
    if (D == 768) ln_row_reg<3>(src, nullptr, w, b, y + (long long)row * D, eps, threadIdx.x & 63, split_out != 0);
    else ln_row(src, nullptr, w, b, y + (long long)row * D, D, eps, threadIdx.x & 63, split_out != 0);
}

// x_pre[b,n,:] = (n == 0 ? class_embedding : patch[b,n-1,:]) + pos[n,:];  x = ln_pre(x_pre)
__global__ __launch_bounds__(256) void assemble_ln_pre_kernel(const float* __restrict__ patch, const float* __restrict__ cls_emb,
                                                              const float* __restrict__ pos, const float* __restrict__ w,
                                                              const float* __restrict__ b, float* __restrict__ x,
                                                              int B, int tokN, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * tokN) return;
    const int bi = row / tokN, n = row % tokN;
    const float* src = (n == 0) ? cls_emb : patch + ((long long)bi * (tokN - 1) + (n - 1)) * D;
    if (D == 768) ln_row_reg<3>(src, pos + (long long)n * D, w, b, x + (long long)row * D, eps, threadIdx.x & 63, false);
    else ln_row(src, pos + (long long)n * D, w, b, x + (long long)row * D, D, eps, threadIdx.x & 63);
}

// column sums of squares over the token axis: ss[b,c] = sum_n f[b,n,c]^2   (64 columns per block)
__global__ __launch_bounds__(256) void token_axis_sumsq_kernel(const float* __restrict__ f, float* __restrict__ ss,
                                                               int tokN, int C) {
    __shared__ float part[4][64];
    const int b = blockIdx.y;
    const int cl = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + cl;
    const int g = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int n = g; n < tokN; n += 4) {
            const float v = f[((long long)b * tokN + n) * C + c];
            s += v * v;
        }
    part[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < C) ss[(long long)b * C + c] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

__global__ __launch_bounds__(256) void token_axis_scale_kernel(const float* __restrict__ f, const float* __restrict__ ss,
                                                               float* __restrict__ out, int tokN, int C, long long total4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const long long e = i * 4;
    const int c = (int)(e % C);
    const long long b = e / ((long long)tokN * C);
    f32x4 v = *reinterpret_cast<const f32x4*>(f + e);
    const f32x4 s = *reinterpret_cast<const f32x4*>(ss + b * C + c);
    v[0] = v[0] / sqrtf(s[0]);
    v[1] = v[1] / sqrtf(s[1]);
    v[2] = v[2] / sqrtf(s[2]);
    v[3] = v[3] / sqrtf(s[3]);
    *reinterpret_cast<f32x4*>(out + e) = v;
}

// im2col for the stride-16 patch conv: column index = c*ps*ps + py*ps + px (conv weight [D,3,ps,ps] flattened)
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, float* __restrict__ col,
                                                     int S, int g, int ps, long long total4, int split_out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const long long e = i * 4;
    const int Kc = 3 * ps * ps;
    const int k = (int)(e % Kc);
    const long long pr = e / Kc;            // b*P + p
    const int P = g * g;
    const int b = (int)(pr / P), pp = (int)(pr % P);
    const int gy = pp / g, gx = pp % g;
    const int c = k / (ps * ps), py = (k / ps) % ps, px = k % ps;   // px multiple of 4
    const f32x4 v = *reinterpret_cast<const f32x4*>(img + (((long long)b * 3 + c) * S + gy * ps + py) * S + gx * ps + px);
    if (split_out) {
        split_t hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { hi[j] = split_hi(v[j]); lo[j] = split_hi(v[j] - (float)hi[j]); }
        split_t* o = reinterpret_cast<split_t*>(col) + pr * 2 * Kc + split_off(k, 0);
        *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(hi);
        *reinterpret_cast<uint2*>(o + 32) = *reinterpret_cast<const uint2*>(lo);
    } else {
        *reinterpret_cast<f32x4*>(col + e) = v;
    }
}

int excel_launch_layernorm(const float* x, const float* cls_src, int tokN, const float* w, const float* b, float* y,
                           int rows, int D, float eps, hipStream_t st, int split_out, long long in_stride) {
    ProfScope prof__(PROF_LAYERNORM, st);
    EXCEL_CHECK_ARG(rows > 0 && D > 0 && (D % 4) == 0, "layernorm: D must be a multiple of 4 (D=%d)", D);
    hipLaunchKernelGGL(layernorm_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, cls_src, tokN, w, b, y, rows, D, eps, split_out, in_stride > 0 ? in_stride : (long long)D);
    EXCEL_CHECK_LAUNCH("layernorm_rows");
    return EXCEL_OK;
}

int excel_launch_assemble_ln_pre(const float* patch, const float* cls_emb, const float* pos, const float* w, const float* b,
                                 float* x, int B, int tokN, int D, float eps, hipStream_t st) {
    ProfScope prof__(PROF_EMBED, st);
    EXCEL_CHECK_ARG((D % 4) == 0, "assemble_ln_pre: D must be a multiple of 4");
    hipLaunchKernelGGL(assemble_ln_pre_kernel, dim3(cdiv(B * tokN, 4)), dim3(256), 0, st, patch, cls_emb, pos, w, b, x, B, tokN, D, eps);
    EXCEL_CHECK_LAUNCH("assemble_ln_pre");
    return EXCEL_OK;
}

int excel_launch_token_axis_normalize(const float* f, float* ss, float* out, int B, int tokN, int C, hipStream_t st) {
    ProfScope prof__(PROF_TOKEN_NORM, st);
    EXCEL_CHECK_ARG((C % 4) == 0, "token_axis_normalize: C must be a multiple of 4");
    hipLaunchKernelGGL(token_axis_sumsq_kernel, dim3(cdiv(C, 64), B), dim3(256), 0, st, f, ss, tokN, C);
    EXCEL_CHECK_LAUNCH("token_axis_sumsq");
    const long long total4 = (long long)B * tokN * C / 4;
    hipLaunchKernelGGL(token_axis_scale_kernel, dim3((unsigned)cdivl(total4, 256)), dim3(256), 0, st, f, ss, out, tokN, C, total4);
    EXCEL_CHECK_LAUNCH("token_axis_scale");
    return EXCEL_OK;
}

int excel_launch_im2col(const float* img, float* col, int B, int S, int ps, hipStream_t st, int split_out) {
    ProfScope prof__(PROF_EMBED, st);
    EXCEL_CHECK_ARG(S % ps == 0 && (ps % 4) == 0, "im2col: S must be a multiple of the patch size, patch %% 4 == 0");
    const int g = S / ps;
    const long long total4 = (long long)B * g * g * 3 * ps * ps / 4;
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)cdivl(total4, 256)), dim3(256), 0, st, img, col, S, g, ps, total4, split_out);
    EXCEL_CHECK_LAUNCH("im2col");
    return EXCEL_OK;
}

}  // namespace EXCEL_SPLIT_NS
