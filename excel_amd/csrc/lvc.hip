// LVC ("learned visual cue") side of the path, SURVEY 8(f) rank 1 -- the pieces around the decoder features:
//   * feature affinity: channel-normalised token similarity, shifted by beta x its mean over the WHOLE batch tensor and
//     scaled by gamma; then either sigmoid (attn_pred, model/model_excel.py:70-76) or "negatives -> -inf, row softmax"
//     (ex_attn, clip/clip_surgery_model.py:128-137);
//   * seg_attn layer selection of refine_cams_with_aff (utils/affutils.py:182-195).
// All reductions are fixed-order (no atomics): results are run-to-run identical.
#include "common.h"
#include "excel_internal.h"

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------- feature affinity
// f [B,C,P]: inv[b,p] = 1 / max(||f[b,:,p]||, 1e-12)        (F.normalize(dim=1))
__global__ __launch_bounds__(256) void lvc_col_invnorm_kernel(const float* __restrict__ f, float* __restrict__ inv, int C, int P) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float* src = f + (long long)b * C * P + p;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) { const float v = src[(long long)c * P]; ss = fmaf(v, v, ss); }
    inv[(long long)b * P + p] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
}

// fn[b,p,c] = f[b,c,p] * inv[b,p]   (row-major [P,Cp] per image, zero-padded to Cp % 4 == 0: the GEMM's K operand)
__global__ __launch_bounds__(256) void lvc_transpose_scale_kernel(const float* __restrict__ f, const float* __restrict__ inv,
                                                                  float* __restrict__ fn, int C, int Cp, int P) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        tile[j][tx] = (c < C && p < P) ? f[((long long)b * C + c) * P + p] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (p < P && c < Cp) fn[((long long)b * P + p) * Cp + c] = tile[tx][j] * inv[(long long)b * P + p];
    }
}

// fixed-order two-stage sum in double: partial[i] = sum of chunk i
__global__ __launch_bounds__(256) void lvc_partial_sum_kernel(const float* __restrict__ x, long long n, double* __restrict__ partial) {
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

__global__ void lvc_final_mean_kernel(const double* __restrict__ partial, int n_partial, long long n, float* __restrict__ mean) {
    if (threadIdx.x || blockIdx.x) return;
    double s = 0.0;
    for (int i = 0; i < n_partial; ++i) s += partial[i];
    mean[0] = (float)(s / (double)n);
}

// one wave per row: z = (sim - mean*beta)*gamma ; mode 0: sigmoid(z) ; mode 1: z < 0 -> -inf, softmax over the row
__global__ __launch_bounds__(256) void lvc_finish_kernel(float* __restrict__ sim, const float* __restrict__ mean, long long rows, int P,
                                                         float beta, float gamma, int mode) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* r = sim + row * P;
    const float shift = mean[0] * beta;
    if (mode == 0) {
        for (int i = lane; i < P; i += 64) {
            const float z = (r[i] - shift) * gamma;
            r[i] = 1.f / (1.f + expf(-z));
        }
        return;
    }
    float m = -INFINITY;
    for (int i = lane; i < P; i += 64) {
        const float z = (r[i] - shift) * gamma;
        if (!(z < 0.f)) m = fmaxf(m, z);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < P; i += 64) {
        const float z = (r[i] - shift) * gamma;
        if (!(z < 0.f)) s += expf(z - m);
    }
    s = wave_sum(s);
    for (int i = lane; i < P; i += 64) {
        const float z = (r[i] - shift) * gamma;
        r[i] = (z < 0.f) ? 0.f : expf(z - m) / s;       // a row of all -inf gives 0/0 = NaN like torch.softmax
    }
}

size_t excel_feature_affinity_ws_bytes(int B, int C, int P) {
    const int Cp = (C + 3) / 4 * 4;
    return align_up((size_t)B * P * sizeof(float), 256) + align_up((size_t)B * P * Cp * sizeof(float), 256) +
           align_up(1024 * sizeof(double), 256) + 256;
}

int excel_launch_feature_affinity(const float* feats, int B, int C, int P, float beta, float gamma, int mode, float* out, void* ws,
                                  hipStream_t st) {
    ProfScope prof__(PROF_OTHER, st);
    EXCEL_CHECK_ARG(feats && out && ws && B > 0 && C > 0 && P > 0 && (mode == 0 || mode == 1), "feature_affinity: bad argument");
    const int Cp = (C + 3) / 4 * 4;
    char* base = (char*)ws;
    float* inv = (float*)base;
    base += align_up((size_t)B * P * sizeof(float), 256);
    float* fn = (float*)base;
    base += align_up((size_t)B * P * Cp * sizeof(float), 256);
    double* partial = (double*)base;
    base += align_up(1024 * sizeof(double), 256);
    float* mean = (float*)base;
    hipLaunchKernelGGL(lvc_col_invnorm_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, st, feats, inv, C, P);
    hipLaunchKernelGGL(lvc_transpose_scale_kernel, dim3(cdiv(P, 32), cdiv(Cp, 32), B), dim3(256), 0, st, feats, inv, fn, C, Cp, P);
    EXCEL_CHECK_LAUNCH("feature_affinity/normalize");
    // sim[b] = fn[b] . fn[b]^T  (exact fp32 MFMA GEMM, NT form, batched)
    GemmArgs g{};
    g.A = fn; g.B = fn; g.C = out; g.bias = nullptr; g.res = nullptr;
    g.M = P; g.N = P; g.K = Cp; g.Kld = Cp; g.lda = Cp; g.ldb = Cp; g.ldc = P; g.ldr = 0;
    g.sA = g.sB = (long long)P * Cp; g.sC = (long long)P * P; g.sR = 0; g.sBias = 0;
    g.zdiv = 1; g.sA2 = g.sB2 = g.sC2 = 0;
    g.act = GEMM_ACT_NONE; g.out_mode = GEMM_OUT_PLAIN; g.tokN = g.heads = g.hd = 0; g.alpha = 1.f;
    int rc = excel_launch_gemm(g, true, B, st);
    if (rc) return rc;
    const long long n = (long long)B * P * P;
    const int nparts = (int)min((long long)1024, cdivl(n, 4096));
    hipLaunchKernelGGL(lvc_partial_sum_kernel, dim3(nparts), dim3(256), 0, st, out, n, partial);
    hipLaunchKernelGGL(lvc_final_mean_kernel, dim3(1), dim3(64), 0, st, partial, nparts, n, mean);
    hipLaunchKernelGGL(lvc_finish_kernel, dim3((unsigned)cdivl((long long)B * P, 4)), dim3(256), 0, st, out, mean, (long long)B * P, P,
                       beta, gamma, mode);
    EXCEL_CHECK_LAUNCH("feature_affinity");
    return EXCEL_OK;
}

// ---------------------------------------------------------------------------------------------- seg_attn layer selection
// attn [Lw,B,N,N] (stacked per-layer maps, the last n_layers of which are used), seg [B,P,P], P = N-1.
// diff[b,l] = sum_{m,n} (seg[b,m,n] - attn[l,b,1+m,1+n]) ; one block per (chunk, l, b), fixed order, double
__global__ __launch_bounds__(256) void lvc_layer_diff_kernel(const float* __restrict__ attn, const float* __restrict__ seg, int B, int N,
                                                             int first_layer, double* __restrict__ partial) {
    const int l = blockIdx.y, b = blockIdx.z, P = N - 1;
    const long long n = (long long)P * P;
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = min(lo + per, n);
    const float* a = attn + (((long long)(first_layer + l) * B + b) * N) * N;
    const float* s = seg + (long long)b * n;
    double acc = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const int m = (int)(i / P), k = (int)(i - (long long)m * P);
        acc += (double)(s[i] - a[(long long)(m + 1) * N + (k + 1)]);
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[((long long)b * gridDim.y + l) * gridDim.x + blockIdx.x] = red[0];
}

// per image: layer mask (diff <= mean diff), 1 / (count + 1e-5)
__global__ void lvc_layer_mask_kernel(const double* __restrict__ partial, int nchunk, int L, float* __restrict__ mask /*[B,L+1]*/) {
    const int b = blockIdx.x;
    if (threadIdx.x) return;
    float diff[16];
    float mean = 0.f;
    for (int l = 0; l < L; ++l) {
        double s = 0.0;
        for (int c = 0; c < nchunk; ++c) s += partial[((long long)b * L + l) * nchunk + c];
        diff[l] = (float)s;
        mean += diff[l];
    }
    mean /= (float)L;                                           // :185
    float cnt = 0.f;
    for (int l = 0; l < L; ++l) {
        const float mk = diff[l] <= mean ? 1.f : 0.f;           // :187-188
        mask[b * (L + 1) + l] = mk;
        cnt += mk;
    }
    mask[b * (L + 1) + L] = 1.f / (cnt + 1e-5f);                // :193
}

__global__ __launch_bounds__(256) void lvc_select_mean_kernel(const float* __restrict__ attn, const float* __restrict__ seg,
                                                              const float* __restrict__ mask, int B, int N, int first_layer, int L,
                                                              float* __restrict__ out) {
    const int P = N - 1;
    const long long n = (long long)P * P;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    const int m = (int)(i / P), k = (int)(i - (long long)m * P);
    float acc = 0.f;
    for (int l = 0; l < L; ++l)
        acc += mask[b * (L + 1) + l] * attn[(((long long)(first_layer + l) * B + b) * N + (m + 1)) * N + (k + 1)];
    out[(long long)b * n + i] = acc * mask[b * (L + 1) + L] * seg[(long long)b * n + i];      // :193, :195
}

size_t excel_attn_select_ws_bytes(int B, int n_layers) {
    return align_up((size_t)B * n_layers * 64 * sizeof(double), 256) + align_up((size_t)B * (n_layers + 1) * sizeof(float), 256);
}

int excel_launch_attn_select_mean(const float* attn, int Lw, int B, int N, int first_layer, int n_layers, const float* seg_attn,
                                  float* out, void* ws, hipStream_t st) {
    ProfScope prof__(PROF_OTHER, st);
    EXCEL_CHECK_ARG(attn && seg_attn && out && ws && first_layer >= 0 && n_layers >= 1 && n_layers <= 16 && first_layer + n_layers <= Lw,
                    "attn_select_mean: bad layer range");
    double* partial = (double*)ws;
    float* mask = (float*)((char*)ws + align_up((size_t)B * n_layers * 64 * sizeof(double), 256));
    hipLaunchKernelGGL(lvc_layer_diff_kernel, dim3(64, n_layers, B), dim3(256), 0, st, attn, seg_attn, B, N, first_layer, partial);
    hipLaunchKernelGGL(lvc_layer_mask_kernel, dim3(B), dim3(64), 0, st, partial, 64, n_layers, mask);
    const long long n = (long long)(N - 1) * (N - 1);
    hipLaunchKernelGGL(lvc_select_mean_kernel, dim3((unsigned)cdivl(n, 256), B), dim3(256), 0, st, attn, seg_attn, mask, B, N, first_layer,
                       n_layers, out);
    EXCEL_CHECK_LAUNCH("attn_select_mean");
    return EXCEL_OK;
}
