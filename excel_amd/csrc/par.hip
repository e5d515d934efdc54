// Pixel-adaptive refinement (utils/PAR.py:26-92) + arg-max labelling (utils/affutils.py:80-89) +
// confusion-matrix accumulation (utils/evaluate.py:9-20).  All HBM-bound streaming kernels.
//
//   par_affinity : guide image -> aff[B,48,H,W]; per pixel: 48 edge-clamped dilated taps per channel, unbiased
//                  std over the taps, -(|I_nb - I|/(std+1e-8)/w1)^2, mean over RGB, softmax over taps
//                  + w2 * softmax(position term)                                   (PAR.py:70-86)
//   par_iterate  : masks'[c] = sum_t masks[c][nb_t] * aff[t]   (Jacobi step, ping-pong buffers; :88-90)
//                  aff is stored as 48 planes so every tap read is a coalesced row segment; algorithmic
//                  traffic per step = (48 + 2C) * H * W * 4 bytes.
//   argmax_label : label = valid_key[argmax_c]                                      (affutils.py:86-87)
//   confusion    : hist[nc*gt + pred] += 1 over gt < nc                             (evaluate.py:10-14)
#include "common.h"
#include "excel_internal.h"

struct ParDil {
    int d[8];
    float pos_sm[64];   // w2 * softmax over taps of the position term (constant vector, PAR.py:83,86)
};

__device__ __constant__ int TAP_DY[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
__device__ __constant__ int TAP_DX[8] = {-1, 0, 1, -1, 1, -1, 0, 1};

template <int ND>
__global__ __launch_bounds__(256) void par_affinity_kernel(const float* __restrict__ img, float* __restrict__ aff, ParDil dl,
                                                           int H, int W, float w1) {
    constexpr int NT = 8 * ND;
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const long long HW = (long long)H * W;
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const float* ch = img + ((long long)b * 3 + c) * HW;
        const float ctr = ch[(long long)y * W + x];
        float nb[NT];
        float sum = 0.f;
#pragma unroll
        for (int di = 0; di < ND; ++di) {
            const int d = dl.d[di];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int yy = min(max(y + TAP_DY[k] * d, 0), H - 1);
                const int xx = min(max(x + TAP_DX[k] * d, 0), W - 1);
                nb[di * 8 + k] = ch[(long long)yy * W + xx];
                sum += nb[di * 8 + k];
            }
        }
        const float mean = sum / (float)NT;
        float var = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) { const float dv = nb[t] - mean; var += dv * dv; }
        const float den = sqrtf(var / (float)(NT - 1)) + 1e-8f;     // unbiased std (torch.std default)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float a = fabsf(nb[t] - ctr) / den / w1;
            acc[t] -= a * a;
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t] = acc[t] / 3.f; m = fmaxf(m, acc[t]); }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t] = __expf(acc[t] - m); s += acc[t]; }
    float* out = aff + (long long)b * NT * HW + (long long)y * W + x;
#pragma unroll
    for (int t = 0; t < NT; ++t) out[(long long)t * HW] = acc[t] / s + dl.pos_sm[t];
}

#define PAR_CCH 8
template <int ND>
__global__ __launch_bounds__(256) void par_iterate_kernel(const float* __restrict__ aff, const float* __restrict__ in,
                                                          float* __restrict__ out, const int* __restrict__ nchan, ParDil dl,
                                                          int Cmax, int H, int W) {
    constexpr int NT = 8 * ND;
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int nch = nchan ? min(nchan[b], Cmax) : Cmax;
    const long long HW = (long long)H * W;
    const float* a = aff + (long long)b * NT * HW + (long long)y * W + x;
    const float* ib = in + (long long)b * Cmax * HW;
    float* ob = out + (long long)b * Cmax * HW + (long long)y * W + x;
    for (int c0 = 0; c0 < nch; c0 += PAR_CCH) {
        float acc[PAR_CCH];
#pragma unroll
        for (int c = 0; c < PAR_CCH; ++c) acc[c] = 0.f;
#pragma unroll
        for (int di = 0; di < ND; ++di) {
            const int d = dl.d[di];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int yy = min(max(y + TAP_DY[k] * d, 0), H - 1);
                const int xx = min(max(x + TAP_DX[k] * d, 0), W - 1);
                const float wgt = a[(long long)(di * 8 + k) * HW];
                const float* src = ib + (long long)yy * W + xx;
#pragma unroll
                for (int c = 0; c < PAR_CCH; ++c)
                    if (c0 + c < nch) acc[c] += src[(long long)(c0 + c) * HW] * wgt;
            }
        }
#pragma unroll
        for (int c = 0; c < PAR_CCH; ++c)
            if (c0 + c < nch) ob[(long long)(c0 + c) * HW] = acc[c];
    }
}

__global__ __launch_bounds__(256) void bilinear_ac_kernel(const float* __restrict__ in, float* __restrict__ out, int planes,
                                                          int h, int w, int H, int W) {
    // F.interpolate(mode='bilinear', align_corners=True) (PAR.py:67)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)planes * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long pl = i / ((long long)W * H);
    const float sy = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* p = in + pl * h * w;
    const float top = (1.f - lx) * p[y0 * w + x0] + lx * p[y0 * w + x1];
    const float bot = (1.f - lx) * p[y1 * w + x0] + lx * p[y1 * w + x1];
    out[i] = (1.f - ly) * top + ly * bot;
}

__global__ __launch_bounds__(256) void argmax_label_kernel(const float* __restrict__ cams, const int* __restrict__ nchan,
                                                           const int* __restrict__ cls_idx, int Smax, int Cmax,
                                                           long long HW, unsigned char* __restrict__ lab8,
                                                           long long* __restrict__ lab64) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const int nch = nchan ? min(nchan[b], Cmax) : Cmax;
    const float* p = cams + (long long)b * Cmax * HW + i;
    float best = p[0];
    int bi = 0;
    for (int c = 1; c < nch; ++c) {
        const float v = p[(long long)c * HW];
        if (v > best) { best = v; bi = c; }       // first maximum wins, like torch.argmax
    }
    // valid_key = [0, cls+1 ...] (affutils.py:168)
    const int key = (bi == 0) ? 0 : (cls_idx ? cls_idx[(long long)b * Smax + bi - 1] + 1 : bi);
    if (lab8) lab8[(long long)b * HW + i] = (unsigned char)key;
    if (lab64) lab64[(long long)b * HW + i] = key;
}

#define CONF_MAXBINS 8192
__global__ __launch_bounds__(256) void confusion_kernel(const unsigned char* __restrict__ gt, const unsigned char* __restrict__ pred,
                                                        long long n, int nc, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int lh[CONF_MAXBINS];
    const int bins = nc * nc;
    for (int i = threadIdx.x; i < bins; i += 256) lh[i] = 0;
    __syncthreads();
    const long long stride = (long long)gridDim.x * 256 * 16;
    for (long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 16; base < n; base += stride) {
        if (base + 16 <= n) {
            const uint4 g4 = *reinterpret_cast<const uint4*>(gt + base);
            const uint4 p4 = *reinterpret_cast<const uint4*>(pred + base);
            const unsigned int gw[4] = {g4.x, g4.y, g4.z, g4.w}, pw[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int g = (gw[k >> 2] >> (8 * (k & 3))) & 255, p = (pw[k >> 2] >> (8 * (k & 3))) & 255;
                if (g < nc && p < nc) atomicAdd(&lh[g * nc + p], 1u);
            }
        } else {
            for (long long j = base; j < n; ++j) {
                const int g = gt[j], p = pred[j];
                if (g < nc && p < nc) atomicAdd(&lh[g * nc + p], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256)
        if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

// ---------------------------------------------------------------- launchers
static int make_dil(const int* dil, int ndil, float w1, float w2, ParDil* out) {
    EXCEL_CHECK_ARG(ndil >= 1 && ndil <= 8, "PAR: 1..8 dilations supported (got %d)", ndil);
    const int nt = 8 * ndil;
    double pos[64], mean = 0.0;
    for (int i = 0; i < ndil; ++i) {
        out->d[i] = dil[i];
        for (int k = 0; k < 8; ++k) {
            const bool diag = (k == 0 || k == 2 || k == 5 || k == 7);
            pos[i * 8 + k] = (double)(float)((diag ? (float)sqrt(2.0) : 1.f) * (float)dil[i]);   // PAR.py:54-62
            mean += pos[i * 8 + k];
        }
    }
    mean /= nt;
    double var = 0.0;
    for (int t = 0; t < nt; ++t) var += (pos[t] - mean) * (pos[t] - mean);
    const double sd = (nt > 1) ? sqrt(var / (nt - 1)) : 0.0;
    double z[64], zm = -1e300, zs = 0.0;
    for (int t = 0; t < nt; ++t) {
        const double a = pos[t] / (sd + 1e-8) / (double)w1;
        z[t] = -(a * a);
        if (z[t] > zm) zm = z[t];
    }
    for (int t = 0; t < nt; ++t) { z[t] = exp(z[t] - zm); zs += z[t]; }
    for (int t = 0; t < 64; ++t) out->pos_sm[t] = (t < nt) ? (float)((double)w2 * (z[t] / zs)) : 0.f;
    return EXCEL_OK;
}

template <int ND>
static void par_aff_launch(const float* img, float* aff, const ParDil& dl, int B, int H, int W, float w1, hipStream_t st) {
    hipLaunchKernelGGL(par_affinity_kernel<ND>, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(256), 0, st, img, aff, dl, H, W, w1);
}
template <int ND>
static void par_it_launch(const float* aff, const float* in, float* out, const int* nchan, const ParDil& dl, int B, int Cmax,
                          int H, int W, hipStream_t st) {
    hipLaunchKernelGGL(par_iterate_kernel<ND>, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(256), 0, st, aff, in, out, nchan, dl, Cmax, H, W);
}

#define ND_SWITCH(nd, CALL)                 \
    switch (nd) {                           \
        case 1: CALL(1); break;             \
        case 2: CALL(2); break;             \
        case 3: CALL(3); break;             \
        case 4: CALL(4); break;             \
        case 5: CALL(5); break;             \
        case 6: CALL(6); break;             \
        case 7: CALL(7); break;             \
        default: CALL(8); break;            \
    }

int excel_launch_par_affinity(const float* img, float* aff, int B, int H, int W, const int* dil, int ndil, float w1, float w2,
                              hipStream_t st) {
    ProfScope prof__(PROF_PAR_AFFINITY, st);
    ParDil dl;
    int rc = make_dil(dil, ndil, w1, w2, &dl);
    if (rc) return rc;
#define CALL(N) par_aff_launch<N>(img, aff, dl, B, H, W, w1, st)
    ND_SWITCH(ndil, CALL)
#undef CALL
    EXCEL_CHECK_LAUNCH("par_affinity");
    return EXCEL_OK;
}

int excel_launch_par_iterate(const float* aff, const float* in, float* out, const int* nchan, int B, int Cmax, int H, int W,
                             const int* dil, int ndil, hipStream_t st) {
    ProfScope prof__(PROF_PAR_ITERATE, st);
    ParDil dl;
    int rc = make_dil(dil, ndil, 0.3f, 0.01f, &dl);
    if (rc) return rc;
#define CALL(N) par_it_launch<N>(aff, in, out, nchan, dl, B, Cmax, H, W, st)
    ND_SWITCH(ndil, CALL)
#undef CALL
    EXCEL_CHECK_LAUNCH("par_iterate");
    return EXCEL_OK;
}

int excel_launch_bilinear_ac(const float* in, float* out, int planes, int h, int w, int H, int W, hipStream_t st) {
    ProfScope prof__(PROF_OTHER, st);
    const long long n = (long long)planes * H * W;
    hipLaunchKernelGGL(bilinear_ac_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, in, out, planes, h, w, H, W);
    EXCEL_CHECK_LAUNCH("bilinear_ac");
    return EXCEL_OK;
}

int excel_launch_argmax_label(const float* cams, const int* nchan, const int* cls_idx, int B, int Smax, int Cmax, long long HW,
                              unsigned char* lab8, long long* lab64, hipStream_t st) {
    ProfScope prof__(PROF_ARGMAX, st);
    hipLaunchKernelGGL(argmax_label_kernel, dim3((unsigned)cdivl(HW, 256), B), dim3(256), 0, st, cams, nchan, cls_idx, Smax, Cmax, HW, lab8, lab64);
    EXCEL_CHECK_LAUNCH("argmax_label");
    return EXCEL_OK;
}

int excel_launch_confusion(const unsigned char* gt, const unsigned char* pred, long long n, int nc, unsigned long long* hist,
                           hipStream_t st) {
    ProfScope prof__(PROF_CONFUSION, st);
    EXCEL_CHECK_ARG(nc >= 1 && nc * nc <= CONF_MAXBINS, "confusion: num_classes %d too large", nc);
    EXCEL_CHECK_ARG((((uintptr_t)gt | (uintptr_t)pred) & 15) == 0, "confusion: gt/pred must be 16-byte aligned");
    const int blocks = (int)((cdivl(n, 256 * 16) < 2048) ? cdivl(n, 256 * 16) : 2048);
    hipLaunchKernelGGL(confusion_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, gt, pred, n, nc, hist);
    EXCEL_CHECK_LAUNCH("confusion");
    return EXCEL_OK;
}
