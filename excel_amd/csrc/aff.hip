// Attention random walk + CAM up-sampling (utils/affutils.py of the reference), everything device-resident:
// the reference's per-class D2H -> OpenCV -> H2D round trips (affutils.py:207-217, :59-66) are replaced by
// one workgroup per (image, present class) doing threshold + 8-connected components + boxes in LDS.
//
//   colsum / sinkhorn_row : compute_trans_mat's 3 x (column-normalise, row-normalise)        (:11-16)
//   symmetrize            : (T + T^T)/2                                                        (:17)
//   [T.T is either the batched fp32 MFMA GEMM (API parity, :19-20) or, on the fused path, never formed:
//    (T.T)(mask.g) == T (T (mask.g)), two mat-vecs per present class instead of a 2 P^3 GEMM]
//   bbox_mask             : scoremap2bbox (:26-53) + box fill (:209-212) -> v = mask .* g
//   matvec                : u = T v  (one wave per matrix row, all present classes at once)
//   cam_minmax_norm       : c -= min ; c /= 1e-7 + max                                         (:72-73)
//   cam_upsample_bkg      : cv2.resize INTER_LINEAR to the label size (:75), bg = 1 - max_c, cat (:165-166)
#include "common.h"
#include "excel_internal.h"

// ---------------------------------------------------------------- Sinkhorn pieces
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ T, float* __restrict__ cs, int P) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const float* Tb = T + (long long)b * P * P;
    // 8 independent partial sums per thread: 8 loads in flight instead of a serial chain of P/4 dependent ones
    // (the grid is only P/64 x B workgroups, so this kernel was latency-, not bandwidth-bound: 1.2 TB/s)
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < P) {
        int i = g;
        for (; i + 28 < P; i += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += Tb[(long long)(i + 4 * u) * P + c];
        }
        for (int u = 0; i < P; i += 4, ++u) a8[u] += Tb[(long long)i * P + c];
    }
    const float s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    part[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < P) cs[(long long)b * P + c] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// t = in[i,:] / cs ; t = t / sum(t) ; out[i,:] = t      (one wave per row; row kept in registers)
#define SK_MAXP 2048
__global__ __launch_bounds__(256) void sinkhorn_row_kernel(const float* __restrict__ in, const float* __restrict__ cs,
                                                           float* __restrict__ out, int P) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= P) return;
    const float* src = in + ((long long)b * P + i) * P;
    const float* c = cs + (long long)b * P;
    float t[SK_MAXP / 64];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < SK_MAXP / 64; ++k) {
        const int j = lane + 64 * k;
        t[k] = (j < P) ? src[j] / c[j] : 0.f;
        s += t[k];
    }
    s = wave_sum(s);
    float* dst = out + ((long long)b * P + i) * P;
#pragma unroll
    for (int k = 0; k < SK_MAXP / 64; ++k) {
        const int j = lane + 64 * k;
        if (j < P) dst[j] = t[k] / s;
    }
}

__global__ __launch_bounds__(256) void symmetrize_kernel(const float* __restrict__ T, float* __restrict__ out, int P) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const float* Tb = T + (long long)b * P * P;
    float* Ob = out + (long long)b * P * P;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    // transposed tile: rows j0.., cols i0..
    for (int k = ty; k < 32; k += 8) {
        const int rr = j0 + k, cc = i0 + tx;
        tile[k][tx] = (rr < P && cc < P) ? Tb[(long long)rr * P + cc] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int rr = i0 + k, cc = j0 + tx;
        if (rr < P && cc < P) Ob[(long long)rr * P + cc] = (Tb[(long long)rr * P + cc] + tile[tx][k]) / 2.f;
    }
}

// ---------------------------------------------------------------- present-class compaction
__global__ void cls_compact_kernel(const float* __restrict__ onehot, int B, int F, int Smax, int* __restrict__ cls_idx,
                                   int* __restrict__ ncls, int* __restrict__ nchan) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = 0;
    for (int f = 0; f < F; ++f)
        if (onehot[(long long)b * F + f] != 0.f) {
            if (n < Smax) cls_idx[(long long)b * Smax + n] = f;
            ++n;
        }
    for (int s = min(n, Smax); s < Smax; ++s) cls_idx[(long long)b * Smax + s] = -1;
    ncls[b] = n;   // caller checks n <= Smax
    if (nchan) nchan[b] = min(n, Smax) + 1;   // + background channel
}

// ---------------------------------------------------------------- scoremap2bbox + box mask
#define BB_MAXP 1024
__global__ __launch_bounds__(256) void bbox_mask_kernel(const float* __restrict__ attr, const int* __restrict__ cls_idx,
                                                        const int* __restrict__ ncls, int g, int F, int Smax, double thre,
                                                        float* __restrict__ v_out, unsigned char* __restrict__ mask_out) {
    __shared__ int lab[BB_MAXP];
    __shared__ float gv[BB_MAXP];
    __shared__ int bx0[BB_MAXP], bx1[BB_MAXP], by0[BB_MAXP], by1[BB_MAXP];
    __shared__ int roots[BB_MAXP];
    __shared__ int s_max, s_nroots, s_changed;
    const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (s >= min(ncls[b], Smax)) return;
    const int cls = cls_idx[(long long)b * Smax + s];
    const int P = g * g;
    if (tid == 0) { s_max = 0; s_nroots = 0; }
    __syncthreads();
    int u8v[BB_MAXP / 256];
#pragma unroll
    for (int k = 0; k < BB_MAXP / 256; ++k) {
        const int p = tid + 256 * k;
        u8v[k] = 0;
        if (p < P) {
            const float x = attr[((long long)b * P + p) * F + cls];
            gv[p] = x;
            u8v[k] = (int)(x * 255.f) & 255;          // (scoremap * 255).astype(np.uint8): truncation (:28)
            atomicMax(&s_max, u8v[k]);
        }
    }
    __syncthreads();
    const int thr = (int)(thre * (double)s_max);       // int(threshold * np.max(img)) in double (:31)
#pragma unroll
    for (int k = 0; k < BB_MAXP / 256; ++k) {
        const int p = tid + 256 * k;
        if (p < P) {
            lab[p] = (u8v[k] > thr) ? p : -1;           // THRESH_BINARY: strictly greater
            bx0[p] = g; bx1[p] = -1; by0[p] = g; by1[p] = -1;
        }
    }
    __syncthreads();
    // 8-connected components by min-label propagation + pointer jumping, to convergence
    for (int iter = 0; iter < 2 * BB_MAXP; ++iter) {   // converges in <= P sweeps; bound is a safety net
        if (tid == 0) s_changed = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BB_MAXP / 256; ++k) {
            const int p = tid + 256 * k;
            if (p < P && lab[p] >= 0) {
                const int y = p / g, x = p % g;
                int best = lab[p];
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int yy = y + dy, xx = x + dx;
                        if (yy >= 0 && yy < g && xx >= 0 && xx < g) {
                            const int l2 = lab[yy * g + xx];
                            if (l2 >= 0 && l2 < best) best = l2;
                        }
                    }
                const int jump = lab[best];
                if (jump >= 0 && jump < best) best = jump;
                if (best < lab[p]) { atomicMin(&lab[p], best); s_changed = 1; }
            }
        }
        __syncthreads();
        if (!s_changed) break;
        __syncthreads();
    }
    // tight box per component root, root list
#pragma unroll
    for (int k = 0; k < BB_MAXP / 256; ++k) {
        const int p = tid + 256 * k;
        if (p < P && lab[p] >= 0) {
            const int root = lab[p], y = p / g, x = p % g;
            atomicMin(&bx0[root], x); atomicMax(&bx1[root], x);
            atomicMin(&by0[root], y); atomicMax(&by1[root], y);
            if (root == p) roots[atomicAdd(&s_nroots, 1)] = p;
        }
    }
    __syncthreads();
    const int nroots = s_nroots;
#pragma unroll
    for (int k = 0; k < BB_MAXP / 256; ++k) {
        const int p = tid + 256 * k;
        if (p < P) {
            const int y = p / g, x = p % g;
            bool in = false;
            for (int i = 0; i < nroots; ++i) {
                const int rt = roots[i];
                // boundingRect -> x1 = min(x + w, W - 1), y1 = min(y + h, H - 1) (:49-50), end-exclusive fill (:212)
                const int x1 = min(bx1[rt] + 1, g - 1), y1 = min(by1[rt] + 1, g - 1);
                in = in || (x >= bx0[rt] && x < x1 && y >= by0[rt] && y < y1);
            }
            v_out[((long long)b * Smax + s) * P + p] = in ? gv[p] : 0.f;
            if (mask_out) mask_out[((long long)b * Smax + s) * P + p] = in ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------- u[b,s,i] = sum_j T[b,i,j] v[b,s,j]
#define MV_S 8
__global__ __launch_bounds__(256) void matvec_kernel(const float* __restrict__ T, const float* __restrict__ v,
                                                     const int* __restrict__ ncls, float* __restrict__ u, int P, int Smax,
                                                     int s0) {
    extern __shared__ float vs[];   // [MV_S][P]
    const int b = blockIdx.y;
    const int ns = min(min(ncls[b], Smax) - s0, MV_S);
    if (ns <= 0) return;
    for (int i = threadIdx.x; i < ns * P; i += 256) vs[i] = v[((long long)b * Smax + s0) * P + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int rr = 0; rr < 4; ++rr) {
        const int i = blockIdx.x * 16 + wave * 4 + rr;
        if (i >= P) break;
        const float* row = T + ((long long)b * P + i) * P;
        float acc[MV_S];
#pragma unroll
        for (int s = 0; s < MV_S; ++s) acc[s] = 0.f;
        for (int j = lane; j < P; j += 64) {
            const float t = row[j];
#pragma unroll
            for (int s = 0; s < MV_S; ++s)
                if (s < ns) acc[s] += t * vs[s * P + j];
        }
#pragma unroll
        for (int s = 0; s < MV_S; ++s) {
            if (s < ns) {
                const float r = wave_sum(acc[s]);
                if (lane == 0) u[((long long)b * Smax + s0 + s) * P + i] = r;
            }
        }
    }
}

// ---------------------------------------------------------------- per-class min-max (scale_cam_image :72-73)
__global__ __launch_bounds__(256) void cam_minmax_norm_kernel(const float* __restrict__ r, const int* __restrict__ ncls,
                                                              float* __restrict__ rn, int P, int Smax) {
    __shared__ float smn[4], smx[4];
    const int s = blockIdx.x, b = blockIdx.y;
    if (s >= min(ncls[b], Smax)) return;
    const float* src = r + ((long long)b * Smax + s) * P;
    float mn = INFINITY, mx = -INFINITY;
    for (int p = threadIdx.x; p < P; p += 256) { const float v = src[p]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    mn = wave_min(mn); mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    const float den = 1e-7f + (mx - mn);
    float* dst = rn + ((long long)b * Smax + s) * P;
    for (int p = threadIdx.x; p < P; p += 256) dst[p] = (src[p] - mn) / den;
}

// ---------------------------------------------------------------- cv2.resize(INTER_LINEAR) + background channel
// OpenCV: fx = (dx + 0.5) * (src/dst) - 0.5 in double; floor; clamp to the edge with weight 0
struct CvTap { int s0, s1; float f; };
__device__ __forceinline__ CvTap cv_linear_tap(int d, int g, int D) {
    const double fd = ((double)d + 0.5) * ((double)g / (double)D) - 0.5;
    CvTap t;
    t.s0 = (int)floor(fd);
    t.f = (float)(fd - t.s0);
    if (t.s0 < 0) { t.f = 0.f; t.s0 = 0; }
    if (t.s0 >= g - 1) { t.f = 0.f; t.s0 = g - 1; }
    t.s1 = min(t.s0 + 1, g - 1);
    return t;
}
// all present classes of one pixel + the background channel (torch.pow(1 - max, 1.), :165); channels above the present ones are
// zeroed on request (nobody on the path reads them: PAR / arg-max stop at nchan)
__device__ __forceinline__ void cam_upsample_px(const float* __restrict__ rn_b, float* __restrict__ out, long long HW, int g, int ns, int Smax,
                                                const CvTap& tx, const CvTap& ty, int zero_unused) {
    float mx = -INFINITY;
    for (int s = 0; s < ns; ++s) {
        const float* m = rn_b + (long long)s * g * g;
        // horizontal pass first; explicit operations: the uniform and the ragged kernel must round alike
        const float top = fmaf(m[ty.s0 * g + tx.s1], tx.f, __fmul_rn(m[ty.s0 * g + tx.s0], 1.f - tx.f));
        const float bot = fmaf(m[ty.s1 * g + tx.s1], tx.f, __fmul_rn(m[ty.s1 * g + tx.s0], 1.f - tx.f));
        const float v = fmaf(bot, ty.f, __fmul_rn(top, 1.f - ty.f));
        out[(long long)(s + 1) * HW] = v;
        mx = fmaxf(mx, v);
    }
    out[0] = 1.f - mx;
    if (zero_unused)
        for (int s = ns; s < Smax; ++s) out[(long long)(s + 1) * HW] = 0.f;
}

__global__ __launch_bounds__(256) void cam_upsample_bkg_kernel(const float* __restrict__ rn, const int* __restrict__ ncls,
                                                               float* __restrict__ cams, int g, int Smax, int H, int W, int zero_unused) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int ns = min(ncls[b], Smax);
    const long long HW = (long long)H * W;
    cam_upsample_px(rn + (long long)b * Smax * g * g, cams + (long long)b * (Smax + 1) * HW + (long long)y * W + x, HW, g, ns, Smax,
                    cv_linear_tap(x, g, W), cv_linear_tap(y, g, H), zero_unused);
}

// ragged: every image up-sampled to its own (H_b, W_b) into the pitched cams [Smax+1 planes per image]
__global__ __launch_bounds__(256) void cam_upsample_bkg_ragged_kernel(const float* __restrict__ rn, const int* __restrict__ ncls,
                                                                      float* __restrict__ cams, int g, int Smax, TileGeo geo, int zero_unused) {
    const Tile t = tile_of<true>(geo);
    const int x = t.x0 + (threadIdx.x & 63);
    if (x >= t.W) return;
    const int ns = min(ncls[t.b], Smax);
    const CvTap tx = cv_linear_tap(x, g, t.W);
    const float* rn_b = rn + (long long)t.b * Smax * g * g;
    float* cb = cams + (long long)(Smax + 1) * t.base + x;
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        const int y = t.y0 + (threadIdx.x >> 6) + 4 * r;
        if (y < t.H) cam_upsample_px(rn_b, cb + (long long)y * t.Wp, t.HW, g, ns, Smax, tx, cv_linear_tap(y, g, t.H), zero_unused);
    }
}

// ---------------------------------------------------------------- launchers
int excel_launch_trans_mat_sym(const float* W, float* T, float* Tsym, float* cs, int B, int P, hipStream_t st) {
    ProfScope prof__(PROF_SINKHORN, st);
    EXCEL_CHECK_ARG(P <= SK_MAXP, "compute_trans_mat: P=%d exceeds %d", P, SK_MAXP);
    const float* src = W;
    for (int round = 0; round < 3; ++round) {
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(P, 64), B), dim3(256), 0, st, src, cs, P);
        hipLaunchKernelGGL(sinkhorn_row_kernel, dim3(cdiv(P, 4), B), dim3(256), 0, st, src, cs, T, P);
        src = T;
    }
    hipLaunchKernelGGL(symmetrize_kernel, dim3(cdiv(P, 32), cdiv(P, 32), B), dim3(256), 0, st, T, Tsym, P);
    EXCEL_CHECK_LAUNCH("trans_mat_sym");
    return EXCEL_OK;
}

int excel_launch_cls_compact(const float* onehot, int B, int F, int Smax, int* cls_idx, int* ncls, int* nchan, hipStream_t st) {
    ProfScope prof__(PROF_OTHER, st);
    hipLaunchKernelGGL(cls_compact_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, onehot, B, F, Smax, cls_idx, ncls, nchan);
    EXCEL_CHECK_LAUNCH("cls_compact");
    return EXCEL_OK;
}

int excel_launch_bbox_mask(const float* attr, const int* cls_idx, const int* ncls, int B, int g, int F, int Smax, double thre,
                           float* v_out, unsigned char* mask_out, hipStream_t st) {
    ProfScope prof__(PROF_BBOX, st);
    EXCEL_CHECK_ARG(g * g <= BB_MAXP, "scoremap2bbox: grid %dx%d exceeds %d cells", g, g, BB_MAXP);
    hipLaunchKernelGGL(bbox_mask_kernel, dim3(Smax, B), dim3(256), 0, st, attr, cls_idx, ncls, g, F, Smax, thre, v_out, mask_out);
    EXCEL_CHECK_LAUNCH("bbox_mask");
    return EXCEL_OK;
}

int excel_launch_matvec(const float* T, const float* v, const int* ncls, float* u, int B, int P, int Smax, hipStream_t st) {
    ProfScope prof__(PROF_MATVEC, st);
    for (int s0 = 0; s0 < Smax; s0 += MV_S) {
        hipLaunchKernelGGL(matvec_kernel, dim3(cdiv(P, 16), B), dim3(256), MV_S * P * sizeof(float), st, T, v, ncls, u, P, Smax, s0);
    }
    EXCEL_CHECK_LAUNCH("matvec");
    return EXCEL_OK;
}

int excel_launch_cam_upsample_bkg(const float* r, const int* ncls, float* rn, float* cams, int B, int g, int Smax, int H, int W,
                                  int zero_unused, hipStream_t st) {
    ProfScope prof__(PROF_UPSAMPLE, st);
    hipLaunchKernelGGL(cam_minmax_norm_kernel, dim3(Smax, B), dim3(256), 0, st, r, ncls, rn, g * g, Smax);
    hipLaunchKernelGGL(cam_upsample_bkg_kernel, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(256), 0, st, rn, ncls, cams, g, Smax, H, W, zero_unused);
    EXCEL_CHECK_LAUNCH("cam_upsample_bkg");
    return EXCEL_OK;
}

int excel_launch_cam_upsample_bkg_ragged(const float* r, const int* ncls, float* rn, float* cams, int g, int Smax, const TileGeo& geo,
                                         int total_tiles, int zero_unused, hipStream_t st) {
    ProfScope prof__(PROF_UPSAMPLE, st);
    hipLaunchKernelGGL(cam_minmax_norm_kernel, dim3(Smax, geo.B), dim3(256), 0, st, r, ncls, rn, g * g, Smax);
    hipLaunchKernelGGL(cam_upsample_bkg_ragged_kernel, dim3(total_tiles), dim3(256), 0, st, rn, ncls, cams, g, Smax, geo, zero_unused);
    EXCEL_CHECK_LAUNCH("cam_upsample_bkg_ragged");
    return EXCEL_OK;
}
