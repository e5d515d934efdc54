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
#include "common.h"
#include "excel_internal.h"

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

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

int excel_launch_train_losses(const float* seg, const float* attn_pred, const unsigned char* pseudo, int B, int nc, int g_h, int g_w, int H, int W,
                              int radius, int ignore, float w_seg, float w_diver, float* losses, float* d_seg, float* d_attn_pred, void* ws,
                              hipStream_t st) {
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
    hipLaunchKernelGGL(tr_aff_reduce_kernel, dim3(np2), dim3(256), 0, st, attn_pred, pseudo, B, g_h, g_w, H, W, radius, ignore, partial);
    hipLaunchKernelGGL(tr_aff_finish_kernel, dim3(1), dim3(64), 0, st, partial, np2, stats, losses);
    hipLaunchKernelGGL(tr_aff_grad_kernel, dim3((unsigned)cdivl(na, 256)), dim3(256), 0, st, pseudo, B, g_h, g_w, H, W, radius, ignore, stats, w_diver,
                       d_attn_pred);
    EXCEL_CHECK_LAUNCH("train_losses");
    return EXCEL_OK;
}
