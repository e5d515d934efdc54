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
