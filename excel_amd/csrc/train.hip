// One training iteration of the decoder head (scripts/train_voc.py:186-220): the losses with their gradients,
// the backward pass of the head, and the optimizer step.  Exact fp32; every reduction is fixed-order (two-stage, double
// partials), so a step is run-to-run reproducible.
//
// Part A (this section): everything outside the network --
//   seg_loss  = get_seg_loss(F.interpolate(seg, (H,W), bilinear), pseudo)            model/losses.py:4-18, train_voc.py:202-203
//   aff_mask  = cams_to_affinity_label(pseudo, get_mask_by_radius(g, g, radius))     utils/camutils.py:438-476
//   div_loss  = get_aff_loss(attn_pred, aff_mask)                                    model/losses.py:20-31
//   loss = w_seg * seg_loss + w_diver * div_loss                                     train_voc.py:215
// and d loss / d seg, d loss / d attn_pred.
#include "decoder_internal.h"

#define TR_NPART 256

// F.interpolate(bilinear, align_corners=False) source coordinates of output index o (size O) in an input of size I
__device__ __forceinline__ void bil_src(int o, int O, int I, int& i0, int& i1, float& l) {
    const float f = fmaxf(((float)I / (float)O) * ((float)o + 0.5f) - 0.5f, 0.f);
    i0 = min((int)f, I - 1);
    i1 = min(i0 + 1, I - 1);
    l = f - (float)i0;
}

// per-pixel cross entropy of the up-sampled logits, split into the background (label 0) and foreground (label 1..nc-1) terms;
// partial[blk] = {bg_ce_sum, fg_ce_sum, bg_count, fg_count}
__global__ __launch_bounds__(256) void tr_seg_ce_kernel(const float* __restrict__ up, const unsigned char* __restrict__ lab, int nc, long long HW,
                                                        long long npix, int ignore, double* __restrict__ partial) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const long long per = (npix + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = min(lo + per, npix);
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const int L = lab[i];
        if (L == ignore || L >= nc) continue;
        const long long b = i / HW, px = i - b * HW;
        const float* p = up + b * nc * HW + px;
        float m = p[0];
        for (int c = 1; c < nc; ++c) m = fmaxf(m, p[(long long)c * HW]);
        float s = 0.f;
        for (int c = 0; c < nc; ++c) s += expf(p[(long long)c * HW] - m);
        const float ce = (m + logf(s)) - p[(long long)L * HW];
        if (L == 0) { acc[0] += (double)ce; acc[2] += 1.0; } else { acc[1] += (double)ce; acc[3] += 1.0; }
    }
    __shared__ double red[4][256];
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[(long long)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// stats[0..3] = sums of the partials; losses[0] = seg_loss = 0.5 * (bg_ce / (bg_n + 1e-6) + fg_ce / (fg_n + 1e-6))
__global__ void tr_seg_finish_kernel(const double* __restrict__ partial, int nparts, float* __restrict__ stats, float* __restrict__ losses) {
    if (threadIdx.x || blockIdx.x) return;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < nparts; ++i)
        for (int k = 0; k < 4; ++k) a[k] += partial[(long long)i * 4 + k];
    const float bg = (float)a[0] / ((float)a[2] + 1e-6f), fg = (float)a[1] / ((float)a[3] + 1e-6f);
    stats[0] = (float)a[2];
    stats[1] = (float)a[3];
    losses[0] = (bg + fg) * 0.5f;
}

// d loss / d up[b,c,px] = w_seg * 0.5 / (n_group + 1e-6) * (softmax_c - [c == L])     (written in place over `up`)
__global__ __launch_bounds__(256) void tr_seg_grad_kernel(float* __restrict__ up, const unsigned char* __restrict__ lab, int nc, long long HW,
                                                          long long npix, int ignore, const float* __restrict__ stats, float w_seg) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int L = lab[i];
    const long long b = i / HW, px = i - b * HW;
    float* p = up + b * nc * HW + px;
    if (L == ignore || L >= nc) {
        for (int c = 0; c < nc; ++c) p[(long long)c * HW] = 0.f;
        return;
    }
    const float w = w_seg * 0.5f / ((L == 0 ? stats[0] : stats[1]) + 1e-6f);
    float m = p[0];
    for (int c = 1; c < nc; ++c) m = fmaxf(m, p[(long long)c * HW]);
    float s = 0.f;
    for (int c = 0; c < nc; ++c) s += expf(p[(long long)c * HW] - m);
    const float inv = 1.f / s;
    for (int c = 0; c < nc; ++c) {
        const float sm = expf(p[(long long)c * HW] - m) * inv;
        p[(long long)c * HW] = w * (sm - (c == L ? 1.f : 0.f));
    }
}

// adjoint of the bilinear up-sampling: one wave per low-resolution cell gathers, in a fixed order, every high-resolution pixel
// whose interpolation stencil touches the cell (weights recomputed exactly as the forward computes them)
__global__ __launch_bounds__(64) void tr_bilinear_adjoint_kernel(const float* __restrict__ dup, float* __restrict__ dseg, int g_h, int g_w, int H,
                                                                 int W) {
    const int cell = blockIdx.x, plane = blockIdx.y;              // plane = b * nc + c
    const int ci = cell / g_w, cj = cell - ci * g_w;
    const float sy = (float)H / (float)g_h, sx = (float)W / (float)g_w;
    const int y_lo = max(0, (int)floorf(((float)ci - 1.f + 0.5f) * sy - 0.5f) - 1), y_hi = min(H - 1, (int)ceilf(((float)ci + 1.f + 0.5f) * sy));
    const int x_lo = max(0, (int)floorf(((float)cj - 1.f + 0.5f) * sx - 0.5f) - 1), x_hi = min(W - 1, (int)ceilf(((float)cj + 1.f + 0.5f) * sx));
    const int nx = x_hi - x_lo + 1, n = (y_hi - y_lo + 1) * nx;
    const float* src = dup + (long long)plane * H * W;
    float acc = 0.f;
    for (int t = threadIdx.x; t < n; t += 64) {
        const int y = y_lo + t / nx, x = x_lo + t % nx;
        int y0, y1, x0, x1;
        float ly, lx;
        bil_src(y, H, g_h, y0, y1, ly);
        bil_src(x, W, g_w, x0, x1, lx);
        const float wy = (y0 == ci ? 1.f - ly : 0.f) + (y1 == ci ? ly : 0.f);
        const float wx = (x0 == cj ? 1.f - lx : 0.f) + (x1 == cj ? lx : 0.f);
        acc += wy * wx * src[(long long)y * W + x];
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) dseg[(long long)plane * g_h * g_w + cell] = acc;
}

// affinity label of cams_to_affinity_label + get_mask_by_radius for entry (m, n) of image b:
//   1: same class, 0: different, 255: outside the radius window or either token ignored
__device__ __forceinline__ int aff_label(const unsigned char* __restrict__ lab, long long Wimg, int g_w, int stride, int m, int n, int radius, int ignore) {
    const int mh = m / g_w, mw = m - mh * g_w, nh = n / g_w, nw = n - nh * g_w;
    const int lm = lab[(long long)(mh * stride) * Wimg + mw * stride], ln = lab[(long long)(nh * stride) * Wimg + nw * stride];   // nearest: src = dst * stride
    if (abs(mh - nh) > radius || abs(mw - nw) > radius) return 255;
    if (lm == ignore || ln == ignore) return 255;
    return lm == ln ? 1 : 0;
}

// partial[blk] = {pos_count, neg_count, sum pos*(1-x), sum neg*x}
__global__ __launch_bounds__(256) void tr_aff_reduce_kernel(const float* __restrict__ ap, const unsigned char* __restrict__ lab, int B, int g_h, int g_w,
                                                            int H, int W, int radius, int ignore, double* __restrict__ partial) {
    const int P = g_h * g_w, stride = H / g_h;
    const long long n = (long long)B * P * P;
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = min(lo + per, n);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const int b = (int)(i / ((long long)P * P));
        const int rem = (int)(i - (long long)b * P * P);
        const int t = aff_label(lab + (long long)b * H * W, W, g_w, stride, rem / P, rem % P, radius, ignore);
        const float x = ap[i];
        if (t == 1) { acc[0] += 1.0; acc[2] += (double)(1.f - x); }
        else if (t == 0) { acc[1] += 1.0; acc[3] += (double)x; }
    }
    __shared__ double red[4][256];
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[(long long)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void tr_aff_finish_kernel(const double* __restrict__ partial, int nparts, float* __restrict__ stats, float* __restrict__ losses) {
    if (threadIdx.x || blockIdx.x) return;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < nparts; ++i)
        for (int k = 0; k < 4; ++k) a[k] += partial[(long long)i * 4 + k];
    const float pc = (float)a[0] + 1.f, ncnt = (float)a[1] + 1.f;                 // losses.py:23,25
    stats[2] = pc;
    stats[3] = ncnt;
    losses[1] = 0.5f * ((float)a[2] / pc) + 0.5f * ((float)a[3] / ncnt);          // :29-32
}

__global__ __launch_bounds__(256) void tr_aff_grad_kernel(const unsigned char* __restrict__ lab, int B, int g_h, int g_w, int H, int W, int radius,
                                                          int ignore, const float* __restrict__ stats, float w_diver, float* __restrict__ dap) {
    const int P = g_h * g_w, stride = H / g_h;
    const long long n = (long long)B * P * P;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = (int)(i / ((long long)P * P));
    const int rem = (int)(i - (long long)b * P * P);
    const int t = aff_label(lab + (long long)b * H * W, W, g_w, stride, rem / P, rem % P, radius, ignore);
    dap[i] = t == 1 ? -0.5f * w_diver / stats[2] : (t == 0 ? 0.5f * w_diver / stats[3] : 0.f);
}

size_t excel_train_losses_ws_bytes(int B, int nc, int H, int W) {
    return align_up((size_t)B * nc * H * W * sizeof(float), 256) + align_up((size_t)TR_NPART * 4 * sizeof(double), 256) + 256;
}

int excel_launch_train_losses(const float* seg, const float* attn_pred, const unsigned char* pseudo, const unsigned char* aff_labels, int B, int nc,
                              int g_h, int g_w, int H, int W, int radius, int ignore, float w_seg, float w_diver, float* losses, float* d_seg,
                              float* d_attn_pred, void* ws, hipStream_t st) {
    if (!aff_labels) aff_labels = pseudo;
    ProfScope prof__(PROF_OTHER, st);
    EXCEL_CHECK_ARG(seg && attn_pred && pseudo && losses && d_seg && d_attn_pred && ws, "train_losses: null argument");
    EXCEL_CHECK_ARG(B > 0 && nc > 1 && g_h > 0 && g_w > 0 && H % g_h == 0 && W % g_w == 0 && H / g_h == W / g_w, "train_losses: label size must be a multiple of the token grid");
    char* base = (char*)ws;
    float* up = (float*)base;
    base += align_up((size_t)B * nc * H * W * sizeof(float), 256);
    double* partial = (double*)base;
    base += align_up((size_t)TR_NPART * 4 * sizeof(double), 256);
    float* stats = (float*)base;                                  // bg_n, fg_n, pos_count, neg_count
    const long long HW = (long long)H * W, npix = (long long)B * HW;
    int rc = excel_launch_bilinear_resize(seg, up, (long long)B * nc, g_h, g_w, H, W, 0, st);       // train_voc.py:202
    if (rc) return rc;
    const int np1 = (int)min((long long)TR_NPART, cdivl(npix, 1024));
    hipLaunchKernelGGL(tr_seg_ce_kernel, dim3(np1), dim3(256), 0, st, up, pseudo, nc, HW, npix, ignore, partial);
    hipLaunchKernelGGL(tr_seg_finish_kernel, dim3(1), dim3(64), 0, st, partial, np1, stats, losses);
    hipLaunchKernelGGL(tr_seg_grad_kernel, dim3((unsigned)cdivl(npix, 256)), dim3(256), 0, st, up, pseudo, nc, HW, npix, ignore, stats, w_seg);
    hipLaunchKernelGGL(tr_bilinear_adjoint_kernel, dim3(g_h * g_w, B * nc), dim3(64), 0, st, up, d_seg, g_h, g_w, H, W);
    const long long na = (long long)B * g_h * g_w * g_h * g_w;
    const int np2 = (int)min((long long)TR_NPART, cdivl(na, 1024));
    hipLaunchKernelGGL(tr_aff_reduce_kernel, dim3(np2), dim3(256), 0, st, attn_pred, aff_labels, B, g_h, g_w, H, W, radius, ignore, partial);
    hipLaunchKernelGGL(tr_aff_finish_kernel, dim3(1), dim3(64), 0, st, partial, np2, stats, losses);
    hipLaunchKernelGGL(tr_aff_grad_kernel, dim3((unsigned)cdivl(na, 256)), dim3(256), 0, st, aff_labels, B, g_h, g_w, H, W, radius, ignore, stats, w_diver,
                       d_attn_pred);
    EXCEL_CHECK_LAUNCH("train_losses");
    return EXCEL_OK;
}

// =====================================================================================================================
// Part B: the decoder head in training mode -- forward keeping what the backward needs, backward producing the gradient of
// every parameter (same struct layout as the weights), AdamW.  Token-major [M = B*P, E] activations throughout.
// (SegFormerHead: model/segformer_head.py:47-77; DecoderTransformer: model/decoder/TransDecoder.py:62-124; attn_pred:
// model/model_excel.py:70-76.)  Dropout2d of the head (segformer_head.py:66,75) is an identity here: the reference trains
// with it (p = 0.1): a counter-based mask (tr_dropout2d_kernel) keeps a step reproducible.

// ---- elementwise / row kernels
__global__ __launch_bounds__(256) void tr_act_kernel(const float* __restrict__ z, float* __restrict__ h, long long n, int gelu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = z[i];
    h[i] = gelu ? v * (1.f / (1.f + expf(-1.702f * v))) : fmaxf(v, 0.f);
}

// dz = dh * act'(z)   (QuickGELU: s(1.702 z) * (1 + 1.702 z (1 - s(1.702 z)));  ReLU: [z > 0]); in place over dh
__global__ __launch_bounds__(256) void tr_act_bwd_kernel(const float* __restrict__ z, float* __restrict__ dh, long long n, int gelu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = z[i];
    if (gelu) {
        const float s = 1.f / (1.f + expf(-1.702f * v));
        dh[i] *= s * (1.f + 1.702f * v * (1.f - s));
    } else {
        dh[i] = v > 0.f ? dh[i] : 0.f;
    }
}

// LayerNorm backward, one wave per row: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) with g = dy * gamma (+ add[row] if given);
// row statistics are kept for the parameter gradients
__global__ __launch_bounds__(256) void tr_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                        const float* __restrict__ add, float* __restrict__ dx, float2* __restrict__ stat, int M,
                                                        int E, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (long long)row * E;
    const float* dr = dy + (long long)row * E;
    float s = 0.f;
    for (int i = lane; i < E; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)E;
    float v = 0.f;
    for (int i = lane; i < E; i += 64) { const float d = xr[i] - mean; v += d * d; }
    const float rstd = rsqrtf(wave_sum(v) / (float)E + eps);
    float a = 0.f, b2 = 0.f;
    for (int i = lane; i < E; i += 64) {
        const float g = dr[i] * gamma[i], xh = (xr[i] - mean) * rstd;
        a += g;
        b2 += g * xh;
    }
    a = wave_sum(a) / (float)E;
    b2 = wave_sum(b2) / (float)E;
    for (int i = lane; i < E; i += 64) {
        const float g = dr[i] * gamma[i], xh = (xr[i] - mean) * rstd;
        dx[(long long)row * E + i] = rstd * (g - a - xh * b2) + (add ? add[(long long)row * E + i] : 0.f);
    }
    if (lane == 0) stat[row] = make_float2(mean, rstd);
}

// dgamma[c] = sum_m dy[m,c] * xhat[m,c], dbeta[c] = sum_m dy[m,c]   (one thread per column, rows in order: reproducible)
__global__ __launch_bounds__(64) void tr_ln_param_grad_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float2* __restrict__ stat,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int E) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= E) return;
    float g = 0.f, b = 0.f;
    for (int m = 0; m < M; ++m) {
        const float2 st = stat[m];
        const float d = dy[(long long)m * E + c];
        g += d * (x[(long long)m * E + c] - st.x) * st.y;
        b += d;
    }
    dgamma[c] = g;
    dbeta[c] = b;
}

// out[c] = sum_m a[m*ld + c]   (bias gradients)
__global__ __launch_bounds__(64) void tr_colsum_kernel(const float* __restrict__ a, float* __restrict__ out, int M, int N, int ld) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= N) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int m = 0;
    for (; m + 3 < M; m += 4) {
        s0 += a[(long long)m * ld + c]; s1 += a[(long long)(m + 1) * ld + c]; s2 += a[(long long)(m + 2) * ld + c]; s3 += a[(long long)(m + 3) * ld + c];
    }
    for (; m < M; ++m) s0 += a[(long long)m * ld + c];
    out[c] = (s0 + s1) + (s2 + s3);
}

// softmax backward over rows: ds = scale * p * (dp - sum_j dp_j p_j); pad columns -> 0.  In place over dp.
__global__ __launch_bounds__(256) void tr_softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, long long rows, int P, int Pp, float scale) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* pr = p + row * Pp;
    float* dr = dp + row * Pp;
    float s = 0.f;
    for (int i = lane; i < P; i += 64) s += pr[i] * dr[i];
    s = wave_sum(s);
    for (int i = lane; i < Pp; i += 64) dr[i] = i < P ? scale * pr[i] * (dr[i] - s) : 0.f;
}

// head-major [B,3,H,P,hd] -> rows [B*P, 3*H*hd]   (inverse of the QKV GEMM's scatter)
__global__ __launch_bounds__(256) void tr_headmajor_to_rows_kernel(const float* __restrict__ hm, float* __restrict__ rows, int B, int H, int P, int hd) {
    const long long n = (long long)B * 3 * H * P * hd;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int E = H * hd;
    const int col = (int)(i % (3 * E));
    const long long row = i / (3 * E);
    const int b = (int)(row / P), nn = (int)(row % P);
    const int t = col / E, hh = (col % E) / hd, d = col % hd;
    rows[i] = hm[((((long long)b * 3 + t) * H + hh) * P + nn) * hd + d];
}

// fn[m,:] = x[m,:] / max(||x[m,:]||, 1e-12)   (F.normalize over the channel axis, one wave per token)
__global__ __launch_bounds__(256) void tr_row_normalize_kernel(const float* __restrict__ x, float* __restrict__ fn, float* __restrict__ inv, int M, int E) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float s = 0.f;
    for (int i = lane; i < E; i += 64) { const float v = x[(long long)row * E + i]; s = fmaf(v, v, s); }
    const float iv = 1.f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int i = lane; i < E; i += 64) fn[(long long)row * E + i] = x[(long long)row * E + i] * iv;
    if (lane == 0) inv[row] = iv;
}

// dx[m,:] += inv[m] * (dfn[m,:] - fn[m,:] * <fn[m,:], dfn[m,:]>)
__global__ __launch_bounds__(256) void tr_row_normalize_bwd_kernel(const float* __restrict__ fn, const float* __restrict__ dfn, const float* __restrict__ inv,
                                                                   float* __restrict__ dx, int M, int E) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float s = 0.f;
    for (int i = lane; i < E; i += 64) s += fn[(long long)row * E + i] * dfn[(long long)row * E + i];
    s = wave_sum(s);
    const float iv = inv[row];
    for (int i = lane; i < E; i += 64) dx[(long long)row * E + i] += iv * (dfn[(long long)row * E + i] - fn[(long long)row * E + i] * s);
}

__global__ __launch_bounds__(256) void tr_sum_kernel(const float* __restrict__ x, long long n, double* __restrict__ partial) {
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = min(lo + per, n);
    double s = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) s += (double)x[i];
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void tr_mean_finish_kernel(const double* __restrict__ partial, int nparts, long long n, float* __restrict__ mean) {
    if (threadIdx.x || blockIdx.x) return;
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += partial[i];
    mean[0] = (float)(s / (double)n);
}

// ap = sigmoid((sim - mean * beta) * gamma), in place
__global__ __launch_bounds__(256) void tr_sigmoid_affine_kernel(float* __restrict__ sim, const float* __restrict__ mean, long long n, float beta, float gamma) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    sim[i] = 1.f / (1.f + expf(-(sim[i] - mean[0] * beta) * gamma));
}

// dz = dap * ap * (1 - ap)
__global__ __launch_bounds__(256) void tr_sigmoid_bwd_kernel(const float* __restrict__ ap, const float* __restrict__ dap, float* __restrict__ dz, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = ap[i];
    dz[i] = dap[i] * a * (1.f - a);
}

// G[b] = gamma * ((dz - beta*mean_dz) + its transpose)      (d sim of the symmetric product fn fn^T, written to out)
__global__ __launch_bounds__(256) void tr_symmetrize_grad_kernel(const float* __restrict__ dz, const float* __restrict__ mean_dz, float* __restrict__ out,
                                                                 int B, int P, float beta, float gamma) {
    const long long n = (long long)B * P * P;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = (int)(i / ((long long)P * P));
    const int rem = (int)(i - (long long)b * P * P);
    const int m = rem / P, k = rem % P;
    const float md = beta * mean_dz[0];
    out[i] = gamma * ((dz[i] - md) + (dz[((long long)b * P + k) * P + m] - md));
}

// AdamW (torch.optim.AdamW, amsgrad off): p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(256) void tr_adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                       long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float pv = p[i] * (1.f - lr * wd);
    const float gv = g[i];
    const float mv = b1 * m[i] + (1.f - b1) * gv;
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pv - (lr / bc1) * (mv / denom);
}

extern "C" int excel_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int step, void* stream) {
    EXCEL_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adamw_step: bad argument");
    // bias corrections in double on the host like torch.optim.AdamW (1 - beta2^t in fp32 loses ~6e-5 relative at small t)
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(tr_adamw_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2s);
    EXCEL_CHECK_LAUNCH("adamw_step");
    return EXCEL_OK;
}

// Dropout2d (segformer_head.py:66,75): whole channels of a sample are zeroed with probability p, the rest scaled by 1/(1-p).
// Counter-based: keep(b, c) is a pure function of (seed, b, c), so forward and backward regenerate the same mask and a step is
// reproducible (the reference draws from torch's generator; no parity is possible or claimed for the mask itself).
__device__ __forceinline__ unsigned tr_hash(unsigned a, unsigned b, unsigned c) {
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 0x165667B1u) * 0xC2B2AE35u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void tr_dropout2d_kernel(float* __restrict__ x, int B, int P, int E, float p, unsigned seed) {
    const long long n = (long long)B * P * E;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % E), b = (int)(i / ((long long)P * E));
    const float u = (float)(tr_hash(seed, (unsigned)b, (unsigned)c) >> 8) * (1.f / 16777216.f);
    x[i] = u < p ? 0.f : x[i] * (1.f / (1.f - p));
}

// ---- workspace of one training iteration
struct TrainWs {
    // forward cache
    float *z1, *h1, *cat, *fts, *inv, *fn, *ap;          // fuse + attn_pred branch
    float *xin, *qkv, *pm, *ao, *x1, *zm, *hq, *xfin;    // per decoder layer (xin, qkv, pm, ao, x1, zm, hq are [nl] arrays)
    // backward scratch
    float *dx, *dt1, *dt4, *dqkvh, *dqkvr, *dpm, *tr, *tp, *dfn, *mean, *segt;
    float2* stat;
    double* partial;
    int Pp, Mp, ncp;
    size_t total;
};

static TrainWs train_ws_layout(const excel_decoder_config& c, int B, int g, char* base) {
    TrainWs w;
    const int P = g * g, E = c.embed, L = c.vit_layers, nl = c.dec_layers, H = c.heads;
    const size_t M = (size_t)B * P;
    w.Pp = (P + 3) / 4 * 4;
    w.Mp = (int)((M + 3) / 4 * 4);
    w.ncp = (c.num_classes + 3) / 4 * 4;
    size_t off = 0;
    auto take = [&](size_t floats) { float* p = (float*)(base + off); off += align_up(floats * sizeof(float), 256); return p; };
    w.z1 = take((size_t)L * M * E); w.h1 = take((size_t)L * M * E); w.cat = take(M * (size_t)L * E); w.fts = take(M * E);
    w.inv = take(M); w.fn = take(M * E); w.ap = take((size_t)B * P * P);
    w.xin = take((size_t)nl * M * E); w.qkv = take((size_t)nl * M * 3 * E); w.pm = take((size_t)nl * B * H * P * w.Pp);
    w.ao = take((size_t)nl * M * E); w.x1 = take((size_t)nl * M * E); w.zm = take((size_t)nl * M * 4 * E); w.hq = take((size_t)nl * M * 4 * E);
    w.xfin = take(M * E);
    w.dx = take(M * E); w.dt1 = take(M * E); w.dt4 = take(M * 4 * E); w.dqkvh = take(M * 3 * E); w.dqkvr = take(M * 3 * E);
    w.dpm = take((size_t)B * H * P * w.Pp);
    const size_t widest = (size_t)(4 * E > L * E ? 4 * E : L * E);
    w.tr = take(widest * (size_t)w.Mp > (size_t)B * H * P * w.Pp ? widest * (size_t)w.Mp : (size_t)B * H * P * w.Pp);
    w.tp = take((size_t)B * H * P * w.Pp);
    w.dfn = take(M * E > (size_t)B * P * P ? M * E : (size_t)B * P * P);
    w.mean = take(64);
    w.segt = take(M * w.ncp);
    w.stat = (float2*)take(2 * M);
    w.partial = (double*)take(2 * TR_NPART);
    w.total = off;
    return w;
}

extern "C" size_t excel_decoder_train_workspace_bytes(excel_decoder_t h, int B, int g) {
    if (!h || B <= 0 || g <= 0) return 0;
    return train_ws_layout(h->cfg, B, g, nullptr).total;
}

static int tr_launch_act(const float* z, float* hh, long long n, int gelu, hipStream_t st) {
    hipLaunchKernelGGL(tr_act_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, z, hh, n, gelu);
    EXCEL_CHECK_LAUNCH("train/act");
    return EXCEL_OK;
}

// forward in training mode: seg [B,nc,g,g], attn_pred [B,P,P]; `workspace` keeps the activations for excel_decoder_backward
extern "C" int excel_decoder_forward_train(excel_decoder_t h, const float* all_feats, int B, int g, void* workspace, size_t workspace_bytes,
                                           float* seg_out, float* attn_pred_out, float dropout_p, unsigned dropout_seed, void* stream) {
    EXCEL_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "excel_decoder_forward_train: dropout_p must be in [0,1)");
    EXCEL_CHECK_ARG(h && all_feats && workspace && seg_out && attn_pred_out && B > 0 && g > 0 && (g * g) % 4 == 0,
                    "excel_decoder_forward_train: bad argument (the token count g*g must be a multiple of 4)");
    const excel_decoder_config& c = h->cfg;
    const int P = g * g, N = P + 1, D = c.vit_width, E = c.embed, L = c.vit_layers, H = c.heads, hd = E / H, nc = c.num_classes, nl = c.dec_layers;
    const int M = B * P;
    hipStream_t st = (hipStream_t)stream;
    TrainWs ws = train_ws_layout(c, B, g, (char*)workspace);
    EXCEL_CHECK_ARG(workspace_bytes >= ws.total, "excel_decoder_forward_train: workspace too small (%zu < %zu)", workspace_bytes, ws.total);
    const size_t ME = (size_t)M * E;
    for (int l = 0; l < L; ++l) {                                                                // segformer_head.py:68-73
        const excel_fuse_layer_weights& fw = h->fuse[l];
        GemmArgs a = ga0(all_feats + ((size_t)l * B * N + 1) * D, fw.proj_w, ws.z1 + l * ME, fw.proj_b, nullptr, P, E, D, D, D, E, 0, GEMM_ACT_NONE);
        a.sA = (long long)N * D; a.sC = (long long)P * E;
        TRYD(excel_launch_gemm(a, true, B, st));
        TRYD(tr_launch_act(ws.z1 + l * ME, ws.h1 + l * ME, (long long)ME, 0, st));
        GemmArgs b2 = ga0(ws.h1 + l * ME, fw.proj2_w, ws.cat + (size_t)l * E, fw.proj2_b, nullptr, M, E, E, E, E, L * E, 0, GEMM_ACT_NONE);
        TRYD(excel_launch_gemm(b2, true, 1, st));
    }
    GemmArgs fz = ga0(ws.cat, h->w.fuse_w, ws.fts, h->w.fuse_b, nullptr, M, E, L * E, L * E, L * E, E, 0, GEMM_ACT_NONE);
    TRYD(excel_launch_gemm(fz, true, 1, st));                                                    // :74
    if (dropout_p > 0.f)                                                                         // :75
        hipLaunchKernelGGL(tr_dropout2d_kernel, dim3((unsigned)cdivl((long long)ME, 256)), dim3(256), 0, st, ws.fts, B, P, E, dropout_p, dropout_seed);
    // attn_pred (model_excel.py:70-76)
    hipLaunchKernelGGL(tr_row_normalize_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, ws.fts, ws.fn, ws.inv, M, E);
    GemmArgs sm = ga0(ws.fn, ws.fn, ws.ap, nullptr, nullptr, P, P, E, E, E, P, 0, GEMM_ACT_NONE);
    sm.sA = sm.sB = (long long)P * E; sm.sC = (long long)P * P;
    TRYD(excel_launch_gemm(sm, true, B, st));
    const long long nap = (long long)B * P * P;
    const int np = (int)min((long long)TR_NPART, cdivl(nap, 1024));
    hipLaunchKernelGGL(tr_sum_kernel, dim3(np), dim3(256), 0, st, ws.ap, nap, ws.partial);
    hipLaunchKernelGGL(tr_mean_finish_kernel, dim3(1), dim3(64), 0, st, ws.partial, np, nap, ws.mean);
    hipLaunchKernelGGL(tr_sigmoid_affine_kernel, dim3((unsigned)cdivl(nap, 256)), dim3(256), 0, st, ws.ap, ws.mean, nap, 1.f, 3.f);
    hipMemcpyAsync(attn_pred_out, ws.ap, sizeof(float) * (size_t)nap, hipMemcpyDeviceToDevice, st);
    // decoder transformer (TransDecoder.py:78-83)
    const float scale = 1.f / sqrtf((float)hd);
    const size_t SP = (size_t)B * H * P * ws.Pp;
    const float* x = ws.fts;
    for (int l = 0; l < nl; ++l) {
        const excel_decoder_block_weights& bw = h->blocks[l];
        float* xin = ws.xin + l * ME;
        if (x != xin) hipMemcpyAsync(xin, x, sizeof(float) * ME, hipMemcpyDeviceToDevice, st);
        float* qkv = ws.qkv + (size_t)l * M * 3 * E;
        float* pm = ws.pm + l * SP;
        float* ao = ws.ao + l * ME;
        float* x1 = ws.x1 + l * ME;
        float* zm = ws.zm + (size_t)l * M * 4 * E;
        float* hq = ws.hq + (size_t)l * M * 4 * E;
        TRYD(excel_launch_layernorm(xin, nullptr, 1, bw.ln1_w, bw.ln1_b, ws.dt1, M, E, 1e-5f, st));
        GemmArgs q = ga0(ws.dt1, bw.in_proj_w, qkv, bw.in_proj_b, nullptr, M, 3 * E, E, E, E, 3 * E, 0, GEMM_ACT_NONE);
        q.out_mode = GEMM_OUT_QKV_HEADMAJOR; q.tokN = P; q.heads = H; q.hd = hd;
        TRYD(excel_launch_gemm(q, true, 1, st));
        GemmArgs sc = ga0(qkv, qkv + (size_t)H * P * hd, pm, nullptr, nullptr, P, P, hd, hd, hd, ws.Pp, 0, GEMM_ACT_NONE);
        sc.alpha = scale; sc.zdiv = H;
        sc.sA = sc.sB = (long long)3 * H * P * hd; sc.sA2 = sc.sB2 = (long long)P * hd;
        sc.sC = (long long)H * P * ws.Pp; sc.sC2 = (long long)P * ws.Pp;
        TRYD(excel_launch_gemm(sc, true, B * H, st));
        TRYD(excel_launch_dec_softmax(pm, (long long)B * H * P, P, ws.Pp, 0, st));
        GemmArgs pv = ga0(pm, qkv + (size_t)2 * H * P * hd, ao, nullptr, nullptr, P, hd, P, ws.Pp, hd, E, 0, GEMM_ACT_NONE);
        pv.Kld = ws.Pp; pv.zdiv = H;
        pv.sA = (long long)H * P * ws.Pp; pv.sA2 = (long long)P * ws.Pp;
        pv.sB = (long long)3 * H * P * hd; pv.sB2 = (long long)P * hd;
        pv.sC = (long long)P * E; pv.sC2 = hd;
        TRYD(excel_launch_gemm(pv, false, B * H, st));
        GemmArgs op = ga0(ao, bw.out_proj_w, x1, bw.out_proj_b, xin, M, E, E, E, E, E, E, GEMM_ACT_NONE);        // x1 = xin + out_proj(ao)
        TRYD(excel_launch_gemm(op, true, 1, st));
        TRYD(excel_launch_layernorm(x1, nullptr, 1, bw.ln2_w, bw.ln2_b, ws.dt1, M, E, 1e-5f, st));
        GemmArgs f1 = ga0(ws.dt1, bw.fc1_w, zm, bw.fc1_b, nullptr, M, 4 * E, E, E, E, 4 * E, 0, GEMM_ACT_NONE);
        TRYD(excel_launch_gemm(f1, true, 1, st));
        TRYD(tr_launch_act(zm, hq, (long long)M * 4 * E, 1, st));
        float* xout = (l + 1 < nl) ? ws.xin + (l + 1) * ME : ws.xfin;
        GemmArgs f2 = ga0(hq, bw.fc2_w, xout, bw.fc2_b, x1, M, E, 4 * E, 4 * E, 4 * E, E, E, GEMM_ACT_NONE);          // x2 = x1 + mlp
        TRYD(excel_launch_gemm(f2, true, 1, st));
        x = xout;
    }
    if (nl == 0) hipMemcpyAsync(ws.xfin, ws.fts, sizeof(float) * ME, hipMemcpyDeviceToDevice, st);
    GemmArgs lp = ga0(ws.xfin, h->w.pred_w, ws.segt, h->w.pred_b, nullptr, M, nc, E, E, E, ws.ncp, 0, GEMM_ACT_NONE);
    TRYD(excel_launch_gemm(lp, true, 1, st));
    TRYD(excel_launch_dec_transpose(ws.segt, seg_out, B, P, nc, ws.ncp, P, st));
    return EXCEL_OK;
}

// dW[out,in] = dY^T X ; db = colsum(dY) ; (optional) dX = dY W.   dY [M,out] pitch ldy, X [M,in] pitch ldx, W [out,in].
static int tr_linear_bwd(const float* dY, int ldy, const float* X, int ldx, const float* W, float* dW, float* db, float* dX, int lddx, int M, int out,
                         int in, TrainWs& ws, hipStream_t st) {
    // dY^T [out, Mp] (zero padded K tail) then NN: dW = dY^T [out,M] . X [M,in]
    TRYD(excel_launch_dec_transpose(dY, ws.tr, 1, M, out, ldy, ws.Mp, st));
    GemmArgs gw = ga0(ws.tr, X, dW, nullptr, nullptr, out, in, M, ws.Mp, ldx, in, 0, GEMM_ACT_NONE);
    gw.Kld = ws.Mp;
    TRYD(excel_launch_gemm(gw, false, 1, st));
    if (db) {
        hipLaunchKernelGGL(tr_colsum_kernel, dim3(cdiv(out, 64)), dim3(64), 0, st, dY, db, M, out, ldy);
        EXCEL_CHECK_LAUNCH("train/colsum");
    }
    if (dX) {                                                      // dX = dY [M,out] . W [out,in]   (NN)
        GemmArgs gx = ga0(dY, W, dX, nullptr, nullptr, M, in, out, ldy, in, lddx, 0, GEMM_ACT_NONE);
        TRYD(excel_launch_gemm(gx, false, 1, st));
    }
    return EXCEL_OK;
}

// Backward of excel_decoder_forward_train (same all_feats / workspace).  `grads` has the layout of the weights; every
// pointer in it is WRITTEN (device memory of the parameter's shape).
// attn_fts of the LAST forward_train on this workspace: the fused, Dropout2d-masked features [B*P, E] -> [B,E,g,g].  The training
// loop's LVC cue is `attn_fts.clone().detach()` of the train-mode forward (scripts/train_voc.py:186-189), i.e. the post-dropout tensor.
extern "C" int excel_decoder_train_attn_fts(excel_decoder_t h, int B, int g, const void* workspace, size_t workspace_bytes, float* attn_fts_out,
                                            void* stream) {
    EXCEL_CHECK_ARG(h && workspace && attn_fts_out && B > 0 && g > 0, "excel_decoder_train_attn_fts: bad argument");
    const excel_decoder_config& c = h->cfg;
    TrainWs ws = train_ws_layout(c, B, g, (char*)workspace);
    EXCEL_CHECK_ARG(workspace_bytes >= ws.total, "excel_decoder_train_attn_fts: workspace too small");
    return excel_launch_dec_transpose(ws.fts, attn_fts_out, B, g * g, c.embed, c.embed, g * g, (hipStream_t)stream);
}

extern "C" int excel_decoder_backward(excel_decoder_t h, const float* all_feats, int B, int g, void* workspace, size_t workspace_bytes,
                                      const float* d_seg, const float* d_attn_pred, const excel_decoder_weights* grads, float dropout_p,
                                      unsigned dropout_seed, void* stream) {
    EXCEL_CHECK_ARG(h && all_feats && workspace && d_seg && grads && grads->fuse && grads->blocks && B > 0 && g > 0, "excel_decoder_backward: bad argument");
    const excel_decoder_config& c = h->cfg;
    const int P = g * g, N = P + 1, D = c.vit_width, E = c.embed, L = c.vit_layers, H = c.heads, hd = E / H, nc = c.num_classes, nl = c.dec_layers;
    const int M = B * P;
    hipStream_t st = (hipStream_t)stream;
    TrainWs ws = train_ws_layout(c, B, g, (char*)workspace);
    EXCEL_CHECK_ARG(workspace_bytes >= ws.total, "excel_decoder_backward: workspace too small");
    EXCEL_CHECK_ARG((D % 4) == 0 && (E % 4) == 0, "excel_decoder_backward: widths must be multiples of 4");
    const size_t ME = (size_t)M * E, SP = (size_t)B * H * P * ws.Pp;
    auto W = [](const float* p) { return const_cast<float*>(p); };
    const float scale = 1.f / sqrtf((float)hd);

    // linear_pred: seg [B,nc,P] -> token-major d_seg_tok [M, ncp]
    hipMemsetAsync(ws.segt, 0, sizeof(float) * (size_t)M * ws.ncp, st);
    TRYD(excel_launch_dec_transpose(d_seg, ws.segt, B, nc, P, P, ws.ncp, st));          // [B,nc,P] -> [B,P,ncp] (pad columns zero)
    TRYD(tr_linear_bwd(ws.segt, ws.ncp, ws.xfin, E, h->w.pred_w, W(grads->pred_w), W(grads->pred_b), ws.dx, E, M, nc, E, ws, st));

    for (int l = nl - 1; l >= 0; --l) {
        const excel_decoder_block_weights& bw = h->blocks[l];
        const excel_decoder_block_weights& gb = grads->blocks[l];
        const float* xin = ws.xin + l * ME;
        const float* qkv = ws.qkv + (size_t)l * M * 3 * E;
        const float* pm = ws.pm + l * SP;
        const float* ao = ws.ao + l * ME;
        const float* x1 = ws.x1 + l * ME;
        const float* zm = ws.zm + (size_t)l * M * 4 * E;
        const float* hq = ws.hq + (size_t)l * M * 4 * E;
        // x2 = x1 + fc2(hq): dhq = dx . W2 ; dW2 = dx^T hq
        TRYD(tr_linear_bwd(ws.dx, E, hq, 4 * E, bw.fc2_w, W(gb.fc2_w), W(gb.fc2_b), ws.dt4, 4 * E, M, E, 4 * E, ws, st));
        hipLaunchKernelGGL(tr_act_bwd_kernel, dim3((unsigned)cdivl((long long)M * 4 * E, 256)), dim3(256), 0, st, zm, ws.dt4, (long long)M * 4 * E, 1);
        // y2 = LN2(x1) recomputed (cheap) into dt1 for dW1
        TRYD(excel_launch_layernorm(x1, nullptr, 1, bw.ln2_w, bw.ln2_b, ws.dt1, M, E, 1e-5f, st));
        TRYD(tr_linear_bwd(ws.dt4, 4 * E, ws.dt1, E, bw.fc1_w, W(gb.fc1_w), W(gb.fc1_b), ws.dfn, E, M, 4 * E, E, ws, st));   // dfn <- dy2
        // dx1 = dx + LN2_bwd(dy2)
        hipLaunchKernelGGL(tr_ln_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, x1, ws.dfn, bw.ln2_w, ws.dx, ws.dt1, ws.stat, M, E, 1e-5f);
        hipLaunchKernelGGL(tr_ln_param_grad_kernel, dim3(cdiv(E, 64)), dim3(64), 0, st, x1, ws.dfn, ws.stat, W(gb.ln2_w), W(gb.ln2_b), M, E);
        // now dt1 = dx1.  x1 = xin + out_proj(ao): dao = dx1 . Wo ; dWo = dx1^T ao
        TRYD(tr_linear_bwd(ws.dt1, E, ao, E, bw.out_proj_w, W(gb.out_proj_w), W(gb.out_proj_b), ws.dx, E, M, E, E, ws, st));       // dx <- dao
        // attention backward per (b,h): dO = dao[:, h*hd:(h+1)*hd]
        //   dP = dO V^T   (batched NT, K = hd)
        GemmArgs gp = ga0(ws.dx, qkv + (size_t)2 * H * P * hd, ws.dpm, nullptr, nullptr, P, P, hd, E, hd, ws.Pp, 0, GEMM_ACT_NONE);
        gp.zdiv = H;
        gp.sA = (long long)P * E; gp.sA2 = hd;
        gp.sB = (long long)3 * H * P * hd; gp.sB2 = (long long)P * hd;
        gp.sC = (long long)H * P * ws.Pp; gp.sC2 = (long long)P * ws.Pp;
        TRYD(excel_launch_gemm(gp, true, B * H, st));
        //   dV = P^T dO : P^T via transpose, then NN with B = dO (pitch E)
        TRYD(excel_launch_dec_transpose(pm, ws.tp, B * H, P, P, ws.Pp, ws.Pp, st));
        float* dqh = ws.dqkvh;
        GemmArgs gv = ga0(ws.tp, ws.dx, dqh + (size_t)2 * H * P * hd, nullptr, nullptr, P, hd, P, ws.Pp, E, hd, 0, GEMM_ACT_NONE);
        gv.Kld = ws.Pp; gv.zdiv = H;
        gv.sA = (long long)H * P * ws.Pp; gv.sA2 = (long long)P * ws.Pp;
        gv.sB = (long long)P * E; gv.sB2 = hd;
        gv.sC = (long long)3 * H * P * hd; gv.sC2 = (long long)P * hd;
        TRYD(excel_launch_gemm(gv, false, B * H, st));
        //   dS = scale * P .* (dP - rowsum(dP .* P))
        hipLaunchKernelGGL(tr_softmax_bwd_kernel, dim3((unsigned)cdivl((long long)B * H * P, 4)), dim3(256), 0, st, pm, ws.dpm, (long long)B * H * P, P,
                           ws.Pp, scale);
        //   dQ = dS K (NN) ; dK = dS^T Q (transpose + NN)
        GemmArgs gq = ga0(ws.dpm, qkv + (size_t)H * P * hd, dqh, nullptr, nullptr, P, hd, P, ws.Pp, hd, hd, 0, GEMM_ACT_NONE);
        gq.Kld = ws.Pp; gq.zdiv = H;
        gq.sA = (long long)H * P * ws.Pp; gq.sA2 = (long long)P * ws.Pp;
        gq.sB = (long long)3 * H * P * hd; gq.sB2 = (long long)P * hd;
        gq.sC = (long long)3 * H * P * hd; gq.sC2 = (long long)P * hd;
        TRYD(excel_launch_gemm(gq, false, B * H, st));
        TRYD(excel_launch_dec_transpose(ws.dpm, ws.tp, B * H, P, P, ws.Pp, ws.Pp, st));
        GemmArgs gk = ga0(ws.tp, qkv, dqh + (size_t)H * P * hd, nullptr, nullptr, P, hd, P, ws.Pp, hd, hd, 0, GEMM_ACT_NONE);
        gk.Kld = ws.Pp; gk.zdiv = H;
        gk.sA = (long long)H * P * ws.Pp; gk.sA2 = (long long)P * ws.Pp;
        gk.sB = (long long)3 * H * P * hd; gk.sB2 = (long long)P * hd;
        gk.sC = (long long)3 * H * P * hd; gk.sC2 = (long long)P * hd;
        TRYD(excel_launch_gemm(gk, false, B * H, st));
        hipLaunchKernelGGL(tr_headmajor_to_rows_kernel, dim3((unsigned)cdivl((long long)M * 3 * E, 256)), dim3(256), 0, st, dqh, ws.dqkvr, B, H, P, hd);
        // y1 = LN1(xin) recomputed into dx (free now) for dW_in; dy1 -> dfn
        TRYD(excel_launch_layernorm(xin, nullptr, 1, bw.ln1_w, bw.ln1_b, ws.dx, M, E, 1e-5f, st));
        TRYD(tr_linear_bwd(ws.dqkvr, 3 * E, ws.dx, E, bw.in_proj_w, W(gb.in_proj_w), W(gb.in_proj_b), ws.dfn, E, M, 3 * E, E, ws, st));
        // dxin = dx1 + LN1_bwd(dy1) -> dx
        hipLaunchKernelGGL(tr_ln_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, xin, ws.dfn, bw.ln1_w, ws.dt1, ws.dx, ws.stat, M, E, 1e-5f);
        hipLaunchKernelGGL(tr_ln_param_grad_kernel, dim3(cdiv(E, 64)), dim3(64), 0, st, xin, ws.dfn, ws.stat, W(gb.ln1_w), W(gb.ln1_b), M, E);
        EXCEL_CHECK_LAUNCH("train/block_bwd");
    }
    // ws.dx = d loss / d fts from the decoder path; add the attn_pred path
    if (d_attn_pred) {
        const long long nap = (long long)B * P * P;
        const int np = (int)min((long long)TR_NPART, cdivl(nap, 1024));
        hipLaunchKernelGGL(tr_sigmoid_bwd_kernel, dim3((unsigned)cdivl(nap, 256)), dim3(256), 0, st, ws.ap, d_attn_pred, ws.dpm, nap);   // dz (dpm is free)
        hipLaunchKernelGGL(tr_sum_kernel, dim3(np), dim3(256), 0, st, ws.dpm, nap, ws.partial);
        hipLaunchKernelGGL(tr_mean_finish_kernel, dim3(1), dim3(64), 0, st, ws.partial, np, nap, ws.mean);
        hipLaunchKernelGGL(tr_symmetrize_grad_kernel, dim3((unsigned)cdivl(nap, 256)), dim3(256), 0, st, ws.dpm, ws.mean, ws.tp, B, P, 1.f, 3.f);
        // dfn[b] = G[b] . fn[b]   (batched NN, K = P)
        GemmArgs gf = ga0(ws.tp, ws.fn, ws.dfn, nullptr, nullptr, P, E, P, P, E, E, 0, GEMM_ACT_NONE);
        gf.sA = (long long)P * P; gf.sB = (long long)P * E; gf.sC = (long long)P * E;
        TRYD(excel_launch_gemm(gf, false, B, st));
        hipLaunchKernelGGL(tr_row_normalize_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, ws.fn, ws.dfn, ws.inv, ws.dx, M, E);
        EXCEL_CHECK_LAUNCH("train/attn_pred_bwd");
    }
    if (dropout_p > 0.f)         // the same mask as the forward (a pure function of seed, image, channel)
        hipLaunchKernelGGL(tr_dropout2d_kernel, dim3((unsigned)cdivl((long long)ME, 256)), dim3(256), 0, st, ws.dx, B, P, E, dropout_p, dropout_seed);
    // linear_fuse: fts = cat . Wf^T + bf  ->  dcat [M, L*E] in dt4?  (dt4 holds M*4E floats; dcat needs M*L*E) -> use tr-free buffers: z1 area is still needed, so dcat goes to hq[0..] (free now)
    float* dcat = ws.hq;                                       // nl*M*4E floats >= M*L*E is not guaranteed: checked below
    EXCEL_CHECK_ARG((size_t)nl * 4 >= (size_t)L, "excel_decoder_backward: scratch for d(cat) too small (need dec_layers*4 >= vit_layers)");
    TRYD(tr_linear_bwd(ws.dx, E, ws.cat, L * E, h->w.fuse_w, W(grads->fuse_w), W(grads->fuse_b), dcat, L * E, M, E, L * E, ws, st));
    for (int l = 0; l < L; ++l) {
        const excel_fuse_layer_weights& fw = h->fuse[l];
        const excel_fuse_layer_weights& gfw = grads->fuse[l];
        // z2_l = h1_l . W2^T + b2 (slice l of cat): dh1 = dz2 . W2 ; dW2 = dz2^T h1
        TRYD(tr_linear_bwd(dcat + (size_t)l * E, L * E, ws.h1 + l * ME, E, fw.proj2_w, W(gfw.proj2_w), W(gfw.proj2_b), ws.dt1, E, M, E, E, ws, st));
        hipLaunchKernelGGL(tr_act_bwd_kernel, dim3((unsigned)cdivl((long long)ME, 256)), dim3(256), 0, st, ws.z1 + l * ME, ws.dt1, (long long)ME, 0);
        // z1_l = tok_l . W1^T + b1 with tok_l = rows 1..P of each image of all_feats[l]: dW1 = dz1^T tok (batched over images, summed)
        // -> gather the token rows once into dt4 ([M, D] needs M*D <= M*4E floats)
        if ((size_t)D > (size_t)4 * E) { excel_set_error("excel_decoder_backward: vit_width > 4*embed not supported by the scratch layout"); return EXCEL_ERR_ARG; }
        hipMemcpy2DAsync(ws.dt4, sizeof(float) * (size_t)P * D, all_feats + ((size_t)l * B * N + 1) * D, sizeof(float) * (size_t)N * D,
                         sizeof(float) * (size_t)P * D, B, hipMemcpyDeviceToDevice, st);
        TRYD(tr_linear_bwd(ws.dt1, E, ws.dt4, D, fw.proj_w, W(gfw.proj_w), W(gfw.proj_b), nullptr, 0, M, E, D, ws, st));
    }
    return EXCEL_OK;
}
