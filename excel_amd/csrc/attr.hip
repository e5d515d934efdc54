// One-time / auxiliary kernels around the hot path:
//   attr_aggregate      TSE text-bank fusion (model/load_attr.py:86-119), once per process
//   bilinear_resize     F.interpolate(mode='bilinear', align_corners=False|True): harness input resize
//                       (tools/infer_lam.py:74) and multi-scale CAM resize (utils/camutils.py:41,54)
//   flip_max_normalize  flip-TTA fuse of cure_attr_map_flip (utils/camutils.py:21-26)
#include "common.h"
#include "excel_internal.h"

#define AA_KMAX 256
// one workgroup per text row t: fg rows get the top-(K-drop) soft attention over the attribute bank added, every
// row is L2-normalised and written as a COLUMN of out [C,T].
__global__ __launch_bounds__(256) void attr_aggregate_kernel(const float* __restrict__ text, const float* __restrict__ bank,
                                                             int F, int T, int C, int K, int drop, float* __restrict__ out) {
    __shared__ float logit[AA_KMAX];
    __shared__ float corr[AA_KMAX];
    __shared__ float red[4];
    extern __shared__ float agg[];   // [C]
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* trow = text + (long long)t * C;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    auto block_max = [&](float v) {
        v = wave_max(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    };
    if (t < F) {
        float lg = 0.f;
        if (tid < K)
            for (int c = 0; c < C; ++c) lg += trow[c] * bank[(long long)c * K + tid];       // fg @ bank (:101)
        if (tid < K) logit[tid] = lg;
        __syncthreads();
        // descending rank (stable): dropped iff rank >= K - drop  (sort + corr[:, -topk:] = -inf + scatter, :102-110)
        bool keep = false;
        if (tid < K) {
            int rank = 0;
            for (int j = 0; j < K; ++j) rank += (logit[j] > lg) || (logit[j] == lg && j < tid);
            keep = rank < K - drop;
        }
        const float m = block_max(keep ? lg : -INFINITY);
        const float e = keep ? __expf(lg - m) : 0.f;
        const float s = block_sum(e);
        if (tid < K) corr[tid] = e / s;                                                        // softmax (:112)
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float a = 0.f;
            for (int k = 0; k < K; ++k) a += corr[k] * bank[(long long)c * K + k];
            agg[c] = a + trow[c];                                                               // corr @ bank^T + fg (:113)
        }
    } else {
        for (int c = tid; c < C; c += 256) agg[c] = trow[c];
    }
    __syncthreads();
    float q = 0.f;
    for (int c = tid; c < C; c += 256) q += agg[c] * agg[c];
    const float nrm = sqrtf(block_sum(q));
    for (int c = tid; c < C; c += 256) out[(long long)c * T + t] = agg[c] / nrm;              // :118
}

// source index / weights of F.interpolate(mode='bilinear'): ATen area_pixel_compute_source_index (align_corners=False:
// scale*(dst+0.5)-0.5 clamped at 0) and the two-stage blend, in explicit operations so that every kernel using it rounds alike
struct BilinearTap { int y0, y1, x0, x1; float ly, lx; };
__device__ __forceinline__ BilinearTap bilinear_tap(int x, int y, int h, int w, int H, int W, int align_corners) {
    float fy, fx;
    if (align_corners) {
        fy = ((H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f) * (float)y;
        fx = ((W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f) * (float)x;
    } else {
        fy = fmaxf(__fsub_rn(__fmul_rn((float)h / (float)H, (float)y + 0.5f), 0.5f), 0.f);
        fx = fmaxf(__fsub_rn(__fmul_rn((float)w / (float)W, (float)x + 0.5f), 0.5f), 0.f);
    }
    BilinearTap t;
    t.y0 = min((int)fy, h - 1); t.x0 = min((int)fx, w - 1);
    t.y1 = min(t.y0 + 1, h - 1); t.x1 = min(t.x0 + 1, w - 1);
    t.ly = fy - (float)t.y0; t.lx = fx - (float)t.x0;
    return t;
}
__device__ __forceinline__ float bilinear_blend(const BilinearTap& t, float p00, float p01, float p10, float p11) {
    const float top = fmaf(t.lx, p01, __fmul_rn(1.f - t.lx, p00));
    const float bot = fmaf(t.lx, p11, __fmul_rn(1.f - t.lx, p10));
    return fmaf(t.ly, bot, __fmul_rn(1.f - t.ly, top));
}

__global__ __launch_bounds__(256) void bilinear_resize_kernel(const float* __restrict__ in, float* __restrict__ out, long long planes,
                                                              int h, int w, int H, int W, int align_corners) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= planes * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long pl = i / ((long long)W * H);
    const BilinearTap t = bilinear_tap(x, y, h, w, H, W, align_corners);
    const float* p = in + pl * h * w;
    out[i] = bilinear_blend(t, p[t.y0 * w + t.x0], p[t.y0 * w + t.x1], p[t.y1 * w + t.x0], p[t.y1 * w + t.x1]);
}

// Ragged input side of the harness: decoded uint8 HWC images of different sizes, packed back to back (image b at byte 3 * loff_b),
// -> transforms.normalize_img (datasets/transforms.py:7-14, double intermediate) -> F.interpolate(bilinear, align_corners=False) to
// S x S (tools/infer_lam.py:74) -> out [B,3,S,S].  The same operations as excel_normalize_img_u8 followed by excel_bilinear_resize
// (same bits), without the full-size fp32 intermediate.  grid (cdiv(S*S,256), 3, B)
__global__ __launch_bounds__(256) void normalize_resize_u8_ragged_kernel(const unsigned char* __restrict__ hwc, float* __restrict__ out,
                                                                         const int* __restrict__ tab, int S, double m0, double m1, double m2,
                                                                         double s0, double s1, double s2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * S) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const int h = tab[EXCEL_RAG_REC * b], w = tab[EXCEL_RAG_REC * b + 1];
    const unsigned char* src = hwc + 3ll * tab[EXCEL_RAG_REC * b + 4] + c;
    const int x = i % S, y = i / S;
    const BilinearTap t = bilinear_tap(x, y, h, w, S, S, 0);
    const double m = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    auto px = [&](int yy, int xx) { return (float)(((double)src[((long long)yy * w + xx) * 3] - m) / sd); };
    out[((long long)b * 3 + c) * S * S + i] = bilinear_blend(t, px(t.y0, t.x0), px(t.y0, t.x1), px(t.y1, t.x0), px(t.y1, t.x1));
}

// attr [2B,P,F] (second half computed from horizontally flipped inputs) -> out [B,P,F]:
//   lam = max(lam[:B], flip_x(lam[B:])) ; lam -= min_hw ; lam /= max_hw + 1e-5        (camutils.py:21-25)
__global__ __launch_bounds__(256) void flip_max_normalize_kernel(const float* __restrict__ attr, float* __restrict__ out, int B,
                                                                 int g, int F) {
    __shared__ float smn[4], smx[4];
    const int f = blockIdx.x, b = blockIdx.y, P = g * g;
    float mn = INFINITY, mx = -INFINITY;
    auto val = [&](int p) {
        const int y = p / g, x = p % g;
        const float a = attr[((long long)b * P + p) * F + f];
        const float c = attr[((long long)(b + B) * P + y * g + (g - 1 - x)) * F + f];
        return fmaxf(a, c);
    };
    for (int p = threadIdx.x; p < P; p += 256) { const float v = val(p); mn = fminf(mn, v); mx = fmaxf(mx, v); }
    mn = wave_min(mn); mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    const float den = (mx - mn) + 1e-5f;
    for (int p = threadIdx.x; p < P; p += 256) out[((long long)b * P + p) * F + f] = (val(p) - mn) / den;
}

// multi-scale / flip fuse of cure-style LAMs (utils/camutils.py:41-61 in its evident intent):
//   maps [2B,P,F] of one scale (second half from flipped inputs) -> bilinear (align_corners=False) to (H,W) ->
//   max(lam, flip_x(lam_flipped)) -> acc[B,F,H,W] (+)= ...        (one thread per output pixel, all in one pass)
__global__ __launch_bounds__(256) void lam_scale_accumulate_kernel(const float* __restrict__ maps, float* __restrict__ acc, int B, int g,
                                                                   int F, int H, int W, int init) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * F * H * W;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int f = (int)((i / ((long long)W * H)) % F), b = (int)(i / ((long long)W * H * F));
    auto sample = [&](int bb, int xx) {   // F.interpolate(bilinear, align_corners=False) of map [g,g] at (y, xx)
        const float fy = fmaxf(((float)g / (float)H) * ((float)y + 0.5f) - 0.5f, 0.f);
        const float fx = fmaxf(((float)g / (float)W) * ((float)xx + 0.5f) - 0.5f, 0.f);
        const int y0 = min((int)fy, g - 1), x0 = min((int)fx, g - 1);
        const int y1 = min(y0 + 1, g - 1), x1 = min(x0 + 1, g - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* m = maps + (long long)bb * g * g * F + f;
        const float top = (1.f - lx) * m[(long long)(y0 * g + x0) * F] + lx * m[(long long)(y0 * g + x1) * F];
        const float bot = (1.f - lx) * m[(long long)(y1 * g + x0) * F] + lx * m[(long long)(y1 * g + x1) * F];
        return (1.f - ly) * top + ly * bot;
    };
    const float v = fmaxf(sample(b, x), sample(b + B, W - 1 - x));     // torch.max(lam[:b], lam[b:].flip(-1))
    acc[i] = init ? v : acc[i] + v;
}

// multi-scale / flip fuse of segmentation logits (tools/infer_seg_voc.py:66-82):
//   segs [2B,nc,h,w] of one scale (second half from flipped inputs) -> bilinear (align_corners=False) to (H,W) ->
//   flip_mean ? (seg + flip_x(seg_flipped)) / 2 : seg          (the reference uses the un-flipped half alone at scale 1.0, :69)
//   -> acc[B,nc,H,W] (+)= ... ; the last call folds the mean over scales (:82) through `scale`
__global__ __launch_bounds__(256) void seg_scale_accumulate_kernel(const float* __restrict__ segs, float* __restrict__ acc, int B, int nc,
                                                                   int h, int w, int H, int W, int flip_mean, int init, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * nc * H * W;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long plane = i / ((long long)W * H);           // b * nc + c
    auto sample = [&](long long pl, int xx) {
        const float fy = fmaxf(((float)h / (float)H) * ((float)y + 0.5f) - 0.5f, 0.f);
        const float fx = fmaxf(((float)w / (float)W) * ((float)xx + 0.5f) - 0.5f, 0.f);
        const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* m = segs + pl * h * w;
        const float top = (1.f - lx) * m[y0 * w + x0] + lx * m[y0 * w + x1];
        const float bot = (1.f - lx) * m[y1 * w + x0] + lx * m[y1 * w + x1];
        return (1.f - ly) * top + ly * bot;
    };
    float v = sample(plane, x);
    if (flip_mean) v = (v + sample(plane + (long long)B * nc, W - 1 - x)) / 2.f;     // (segs[:1] + segs[1:].flip(-1)) / 2  (:79)
    v = init ? v : acc[i] + v;
    acc[i] = v * scale;
}

int excel_launch_seg_scale_accumulate(const float* segs, float* acc, int B, int nc, int h, int w, int H, int W, int flip_mean, int init,
                                      float scale, hipStream_t st) {
    const long long total = (long long)B * nc * H * W;
    hipLaunchKernelGGL(seg_scale_accumulate_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, segs, acc, B, nc, h, w, H, W,
                       flip_mean, init, scale);
    EXCEL_CHECK_LAUNCH("seg_scale_accumulate");
    return EXCEL_OK;
}

// denormalize_img / denormalize_img2 (utils/imutils.py:11-25): v = img * std[c] + mean[c] -> truncate to uint8 (torch's
// float -> uint8 cast: toward zero; values are inside [0,255] for real images) ; img2 = that / 255 as float
__global__ __launch_bounds__(256) void denormalize_kernel(const float* __restrict__ img, unsigned char* __restrict__ out8, float* __restrict__ outf,
                                                          long long HW, long long total, float m0, float m1, float m2, float s0, float s1, float s2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / HW) % 3);
    const float v = img[i] * (c == 0 ? s0 : (c == 1 ? s1 : s2)) + (c == 0 ? m0 : (c == 1 ? m1 : m2));
    const unsigned char q = (unsigned char)(int)fminf(fmaxf(v, 0.f), 255.f);
    if (out8) out8[i] = q;
    if (outf) outf[i] = (float)q / 255.0f;
}

// lam_to_label (utils/camutils.py:123-145): valid = cls_label * cam; (value, arg) = max over classes (first maximum); label = arg + 1;
// ignore_mid: value <= high -> ignore, then value <= low -> 0 ; else value <= bkg -> 0.  img_box [B,4] (y0,y1,x0,x1): outside -> ignore.
__global__ __launch_bounds__(256) void lam_to_label_kernel(const float* __restrict__ cam, const float* __restrict__ cls, const int* __restrict__ box,
                                                           int B, int F, int H, int W, float bkg, float high, float low, int ignore_mid, int ignore,
                                                           float* __restrict__ valid, unsigned char* __restrict__ lab) {
    const long long HW = (long long)H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * HW) return;
    const int b = (int)(i / HW);
    const long long px = i - (long long)b * HW;
    float best = 0.f;
    int bi = 0;
    for (int f = 0; f < F; ++f) {
        const float v = cls[(long long)b * F + f] * cam[((long long)b * F + f) * HW + px];
        if (valid) valid[((long long)b * F + f) * HW + px] = v;
        if (f == 0 || v > best) { best = v; bi = f; }
    }
    int l = bi + 1;
    if (ignore_mid) {
        if (best <= high) l = ignore;
        if (best <= low) l = 0;
    } else if (best <= bkg) {
        l = 0;
    }
    if (box) {
        const int y = (int)(px / W), x = (int)(px % W);
        const int* bx = box + b * 4;
        if (!(y >= bx[0] && y < bx[1] && x >= bx[2] && x < bx[3])) l = ignore;
    }
    lab[i] = (unsigned char)l;
}

int excel_launch_lam_to_label(const float* cam, const float* cls, const int* box, int B, int F, int H, int W, float bkg, float high, float low,
                              int ignore_mid, int ignore, float* valid, unsigned char* lab, hipStream_t st) {
    hipLaunchKernelGGL(lam_to_label_kernel, dim3((unsigned)cdivl((long long)B * H * W, 256)), dim3(256), 0, st, cam, cls, box, B, F, H, W, bkg, high, low,
                       ignore_mid, ignore, valid, lab);
    EXCEL_CHECK_LAUNCH("lam_to_label");
    return EXCEL_OK;
}

// transforms.normalize_img + HWC->CHW (datasets/transforms.py, datasets/voc.py:115-116): u8 [B,H,W,3] -> f32 [B,3,H,W],
// (u8 - mean[c]) / std[c] evaluated in double and rounded once, exactly like numpy's float64 intermediate
__global__ __launch_bounds__(256) void normalize_u8_kernel(const unsigned char* __restrict__ hwc, float* __restrict__ out, long long HW, long long total,
                                                           double m0, double m1, double m2, double s0, double s1, double s2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // output index (b, c, pix)
    if (i >= total) return;
    const long long pix = i % HW;
    const int c = (int)((i / HW) % 3);
    const long long b = i / (3 * HW);
    const double v = (double)hwc[(b * HW + pix) * 3 + c];
    out[i] = (float)((v - (c == 0 ? m0 : (c == 1 ? m1 : m2))) / (c == 0 ? s0 : (c == 1 ? s1 : s2)));
}

int excel_launch_normalize_u8(const unsigned char* hwc, float* out, int B, long long HW, const double* mean, const double* stdv, hipStream_t st) {
    const long long total = (long long)B * 3 * HW;
    hipLaunchKernelGGL(normalize_u8_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, hwc, out, HW, total, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    EXCEL_CHECK_LAUNCH("normalize_u8");
    return EXCEL_OK;
}

int excel_launch_denormalize(const float* img, unsigned char* out8, float* outf, int B, long long HW, const float* mean, const float* stdv,
                             hipStream_t st) {
    const long long total = (long long)B * 3 * HW;
    hipLaunchKernelGGL(denormalize_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, img, out8, outf, HW, total, mean[0], mean[1], mean[2],
                       stdv[0], stdv[1], stdv[2]);
    EXCEL_CHECK_LAUNCH("denormalize");
    return EXCEL_OK;
}

// lam = lam - min_hw ; lam /= max_hw + 1e-5 per (b, f) plane (camutils.py:58-59), in place
__global__ __launch_bounds__(256) void plane_minmax_normalize_kernel(float* __restrict__ lam, long long HW) {
    __shared__ float smn[4], smx[4];
    float* pl = lam + (long long)blockIdx.x * HW;
    float mn = INFINITY, mx = -INFINITY;
    for (long long p = threadIdx.x; p < HW; p += 256) { const float v = pl[p]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    mn = wave_min(mn); mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    const float den = (mx - mn) + 1e-5f;
    for (long long p = threadIdx.x; p < HW; p += 256) pl[p] = (pl[p] - mn) / den;
}

int excel_launch_lam_scale_accumulate(const float* maps, float* acc, int B, int g, int F, int H, int W, int init, hipStream_t st) {
    const long long total = (long long)B * F * H * W;
    hipLaunchKernelGGL(lam_scale_accumulate_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, maps, acc, B, g, F, H, W, init);
    EXCEL_CHECK_LAUNCH("lam_scale_accumulate");
    return EXCEL_OK;
}

int excel_launch_plane_minmax_normalize(float* lam, long long planes, long long HW, hipStream_t st) {
    hipLaunchKernelGGL(plane_minmax_normalize_kernel, dim3((unsigned)planes), dim3(256), 0, st, lam, HW);
    EXCEL_CHECK_LAUNCH("plane_minmax_normalize");
    return EXCEL_OK;
}

int excel_launch_attr_aggregate(const float* text, const float* bank, int F, int T, int C, int K, int drop, float* out,
                                hipStream_t st) {
    EXCEL_CHECK_ARG(K >= 1 && K <= AA_KMAX && drop >= 0 && drop < K && F <= T, "attr_aggregate: need K <= %d, 0 <= drop < K", AA_KMAX);
    hipLaunchKernelGGL(attr_aggregate_kernel, dim3(T), dim3(256), C * sizeof(float), st, text, bank, F, T, C, K, drop, out);
    EXCEL_CHECK_LAUNCH("attr_aggregate");
    return EXCEL_OK;
}

int excel_launch_bilinear_resize(const float* in, float* out, long long planes, int h, int w, int H, int W, int align_corners,
                                 hipStream_t st) {
    const long long n = planes * H * W;
    hipLaunchKernelGGL(bilinear_resize_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, in, out, planes, h, w, H, W, align_corners);
    EXCEL_CHECK_LAUNCH("bilinear_resize");
    return EXCEL_OK;
}

int excel_launch_normalize_resize_u8_ragged(const unsigned char* hwc, float* out, const TileGeo& geo, int S, const double* mean, const double* stdv,
                                            hipStream_t st) {
    ProfScope prof__(PROF_OTHER, st);
    hipLaunchKernelGGL(normalize_resize_u8_ragged_kernel, dim3(cdiv(S * S, 256), 3, geo.B), dim3(256), 0, st, hwc, out, geo.tab, S, mean[0], mean[1],
                       mean[2], stdv[0], stdv[1], stdv[2]);
    EXCEL_CHECK_LAUNCH("normalize_resize_u8_ragged");
    return EXCEL_OK;
}

int excel_launch_flip_max_normalize(const float* attr, float* out, int B, int g, int F, hipStream_t st) {
    hipLaunchKernelGGL(flip_max_normalize_kernel, dim3(F, B), dim3(256), 0, st, attr, out, B, g, F);
    EXCEL_CHECK_LAUNCH("flip_max_normalize");
    return EXCEL_OK;
}
