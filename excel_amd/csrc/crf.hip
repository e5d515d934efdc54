// Fully-connected CRF post-processing (utils/dcrf.py:42-68 class DenseCRF, :7-40 crf_inference*; driven by tools/infer_lam.py:179-237):
// mean-field inference with a Gaussian (x,y) and a bilateral (x,y,r,g,b) Potts term, message passing = high-dimensional Gaussian
// filtering on the permutohedral lattice (Adams, Baek & Davis 2010), as Kraehenbuehl & Koltun's densecrf - the library behind the
// reference's pydensecrf dependency - does it.  Everything on the device, one image per call:
//   crf_lattice_kernel<D>   per pixel: feature -> elevate -> nearest 0-coloured lattice point -> rank -> barycentric weights + the D+1 vertex keys
//   crf_hash_insert         open-addressing hash of vertex keys (atomicCAS claims a slot for the first vertex with a key; later ones compare)
//   crf_offsets / crf_neighbors   vertex -> lattice point index; per lattice point and axis the two blur neighbours (key -+ 1, axis j: +- D)
//   splat (64-bit fixed-point atomics: the sum is order independent, so results are bit-reproducible although lattice indices are
//   handed out by an atomic counter) -> D+1 blur passes -> slice, symmetric normalisation 1/sqrt(K 1)
//   mean field: Q = softmax(-U); 10 x { Q = softmax(-U + w_g K_g Q + w_b K_b Q) }
#include "common.h"
#include "excel_internal.h"

#define TRY(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)
#define CRF_KS 8                 // shorts per stored key (D <= 7)
#define CRF_FIX 1099511627776.0  // 2^40 fixed-point scale of the splat accumulators

struct CrfKey { unsigned long long a, b; };   // 8 shorts

__device__ __forceinline__ unsigned crf_hash(const CrfKey& k) {
    unsigned long long h = k.a * 0x9E3779B97F4A7C15ull ^ (k.b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return (unsigned)h;
}
__device__ __forceinline__ void crf_setk(CrfKey& k, int i, int v) {
    const unsigned long long m = (unsigned long long)(unsigned short)(short)v << (16 * (i & 3));
    if (i < 4) k.a |= m; else k.b |= m;
}
__device__ __forceinline__ int crf_getk(const CrfKey& k, int i) {
    return (short)(unsigned short)(((i < 4) ? k.a : k.b) >> (16 * (i & 3)));
}

// feature layout of DenseCRF2D (densecrf.cpp addPairwiseGaussian / addPairwiseBilateral): (x/sxy, y/sxy[, r/srgb, g/srgb, b/srgb])
template <int D>
__global__ __launch_bounds__(256) void crf_lattice_kernel(const unsigned char* __restrict__ rgb, int H, int W, float sxy, float srgb,
                                                          CrfKey* __restrict__ keys, float* __restrict__ bary) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= H * W) return;
    const int x = n % W, y = n / W;
    float f[D];
    f[0] = (float)x / sxy;
    f[1] = (float)y / sxy;
    if (D == 5) {
#pragma unroll
        for (int c = 0; c < 3; ++c) f[2 + c] = (float)rgb[(long long)n * 3 + c] / srgb;
    }
    const float inv_std = sqrtf(2.0f / 3.0f) * (float)(D + 1);      // expected std of the filter (Adams et al. p.6)
    float elevated[D + 1];
    float sm = 0.f;
#pragma unroll
    for (int j = D; j > 0; --j) {
        const float scale = (float)(1.0 / sqrt((double)((j + 1) * j))) * inv_std;
        const float cf = __fmul_rn(f[j - 1], scale);
        elevated[j] = __fsub_rn(sm, __fmul_rn((float)j, cf));
        sm = __fadd_rn(sm, cf);
    }
    elevated[0] = sm;
    const float down = 1.0f / (float)(D + 1), up = (float)(D + 1);
    int rem0[D + 1], rank[D + 1], sum = 0;
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        const float v = __fmul_rn(down, elevated[i]);
        const float u = ceilf(v) * up, dn = floorf(v) * up;
        rem0[i] = (u - elevated[i] < elevated[i] - dn) ? (int)u : (int)dn;
        sum += rem0[i] / (D + 1);
        rank[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i + 1; j <= D; ++j) {
            if (elevated[i] - (float)rem0[i] < elevated[j] - (float)rem0[j]) ++rank[i]; else ++rank[j];
        }
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        rank[i] += sum;
        if (rank[i] < 0) { rank[i] += D + 1; rem0[i] += D + 1; }
        else if (rank[i] > D) { rank[i] -= D + 1; rem0[i] -= D + 1; }
    }
    float bc[D + 2];
#pragma unroll
    for (int i = 0; i <= D + 1; ++i) bc[i] = 0.f;
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        const float v = __fmul_rn(elevated[i] - (float)rem0[i], down);
#pragma unroll
        for (int q = 0; q <= D + 1; ++q) {       // static indexing (a dynamically indexed register array goes to scratch)
            if (q == D - rank[i]) bc[q] += v;
            if (q == D - rank[i] + 1) bc[q] -= v;
        }
    }
    bc[0] += 1.0f + bc[D + 1];
#pragma unroll
    for (int r = 0; r <= D; ++r) {
        CrfKey k{0ull, 0ull};
#pragma unroll
        for (int i = 0; i < D; ++i) {
            // canonical[r][rank] = r for rank <= D - r, else r - (D+1)
            const int can = (rank[i] <= D - r) ? r : r - (D + 1);
            crf_setk(k, i, rem0[i] + can);
        }
        keys[(long long)n * (D + 1) + r] = k;
        bary[(long long)n * (D + 1) + r] = bc[r];
    }
}

__global__ __launch_bounds__(256) void crf_hash_insert_kernel(const CrfKey* __restrict__ keys, long long npv, int* __restrict__ table, unsigned mask,
                                                              int* __restrict__ rep, int* __restrict__ latidx, int* __restrict__ counter,
                                                              CrfKey* __restrict__ lkeys) {
    const long long pv = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pv >= npv) return;
    const CrfKey k = keys[pv];
    unsigned h = crf_hash(k) & mask;
    while (true) {
        const int e = atomicCAS(&table[h], -1, (int)pv);
        if (e == -1) {
            const int li = atomicAdd(counter, 1);
            latidx[pv] = li;
            lkeys[li] = k;
            rep[pv] = (int)pv;
            return;
        }
        const CrfKey ke = keys[e];
        if (ke.a == k.a && ke.b == k.b) { rep[pv] = e; return; }
        h = (h + 1) & mask;
    }
}

__global__ __launch_bounds__(256) void crf_offsets_kernel(const int* __restrict__ rep, const int* __restrict__ latidx, long long npv, int* __restrict__ offset) {
    const long long pv = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pv < npv) offset[pv] = latidx[rep[pv]];
}

__device__ __forceinline__ int crf_find(const CrfKey& k, const CrfKey* keys, const int* table, unsigned mask, const int* latidx) {
    unsigned h = crf_hash(k) & mask;
    while (true) {
        const int e = table[h];
        if (e == -1) return -1;
        const CrfKey ke = keys[e];
        if (ke.a == k.a && ke.b == k.b) return latidx[e];
        h = (h + 1) & mask;
    }
}

template <int D>
__global__ __launch_bounds__(256) void crf_neighbors_kernel(const CrfKey* __restrict__ lkeys, const int* __restrict__ counter, const CrfKey* __restrict__ keys,
                                                            const int* __restrict__ table, unsigned mask, const int* __restrict__ latidx,
                                                            int2* __restrict__ nbr, long long Mcap) {
    const int M = *counter;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int i = (int)(t % Mcap), j = (int)(t / Mcap);
    if (j > D || i >= M) return;
    const CrfKey k = lkeys[i];
    CrfKey n1{0ull, 0ull}, n2{0ull, 0ull};
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const int v = crf_getk(k, c);
        crf_setk(n1, c, (c == j) ? v + D : v - 1);
        crf_setk(n2, c, (c == j) ? v - D : v + 1);
    }
    nbr[(long long)j * Mcap + i] = make_int2(crf_find(n1, keys, table, mask, latidx), crf_find(n2, keys, table, mask, latidx));
}

// ---- message passing of one mean-field step: BOTH kernels (Gaussian: lattice g, bilateral: lattice b) ride in the same launches
//   splat (1 launch) -> blur passes (max(D_g, D_b) + 1 launches; pass 0 reads the fixed-point accumulators, pass 1 re-zeroes them for
//   the next step) -> slice + mean-field update (1 launch): 8 launches per step instead of 18 (a memset, splat, fixed->float, D+1 blurs
//   and a slice per kernel, then the update).  Same operations in the same order as the per-kernel form: same bits.
struct CrfSide {                 // one lattice as the message-passing kernels see it
    const int* offset;           // [npv] vertex -> lattice point
    const float* bary;           // [npv]
    const float* norm;           // [N] symmetric normaliser (may be null while it is being built)
    const int2* nbr;             // [D+1][npv] blur neighbours
    const int* counter;          // number of lattice points
    long long* acc;              // [npv*C] fixed-point splat accumulators
    float *lat0, *lat1;          // [npv*C] ping-pong
    long long npv;
    int Dp1;
    float alpha;                 // 1 / (1 + 2^-D)
};

// splat: acc[o][k] += w * (in[n][k] * norm[n]) in 2^-40 fixed point (integer sums commute: deterministic).  Grid-stride over one side.
__device__ __forceinline__ void crf_splat_side(const float* __restrict__ in, int use_norm, const CrfSide& L, int C) {
    const long long total = L.npv * C, stride = (long long)gridDim.x * 256;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const long long pv = t / C;
        const int k = (int)(t - pv * C);
        const long long n = pv / L.Dp1;
        float v = in[n * C + k];
        if (use_norm) v = __fmul_rn(v, L.norm[n]);
        const float wv = __fmul_rn(L.bary[pv], v);
        atomicAdd(reinterpret_cast<unsigned long long*>(&L.acc[(long long)L.offset[pv] * C + k]), (unsigned long long)(long long)__double2ll_rn((double)wv * CRF_FIX));
    }
}
__global__ __launch_bounds__(256) void crf_splat2_kernel(const float* __restrict__ in, int use_norm, CrfSide g, CrfSide b, int C) {
    crf_splat_side(in, use_norm, g, C);
    crf_splat_side(in, use_norm, b, C);
}
__device__ __forceinline__ float crf_fix(long long a) { return (float)((double)a * (1.0 / CRF_FIX)); }
// blur pass j of one lattice: grid-stride over its M lattice points (M is read on the device: the bilateral lattice of
// tools/infer_lam.py's parameters (sxy 67) has a few thousand points, the launch capacity would be 6 N).  Pass 0 converts the
// accumulators on the fly, pass 1 clears them (nobody reads them after pass 0).
__device__ __forceinline__ void crf_blur_side(const CrfSide& L, int j, int C) {
    if (j >= L.Dp1) return;
    const long long total = (long long)(*L.counter) * C, stride = (long long)gridDim.x * 256;
    float* nw = (j & 1) ? L.lat0 : L.lat1;             // pass 0: acc -> lat1, pass 1: lat1 -> lat0, ...
    const float* old = (j & 1) ? L.lat1 : L.lat0;
    const int2* nbr = L.nbr + (long long)j * L.npv;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int i = (int)(t / C), k = (int)(t - (long long)i * C);
        const int2 nb = nbr[i];
        if (j == 0) {
            const float a = nb.x >= 0 ? crf_fix(L.acc[(long long)nb.x * C + k]) : 0.f, c = nb.y >= 0 ? crf_fix(L.acc[(long long)nb.y * C + k]) : 0.f;
            nw[t] = crf_fix(L.acc[t]) + 0.5f * (a + c);
        } else {
            const float a = nb.x >= 0 ? old[(long long)nb.x * C + k] : 0.f, c = nb.y >= 0 ? old[(long long)nb.y * C + k] : 0.f;
            nw[t] = old[t] + 0.5f * (a + c);
            if (j == 1) L.acc[t] = 0;
        }
    }
}
__global__ __launch_bounds__(256) void crf_blur2_kernel(CrfSide g, CrfSide b, int j, int C) {
    crf_blur_side(g, j, C);
    crf_blur_side(b, j, C);
}
// the Dp1 vertices of pixel n (lattice point, barycentric weight x alpha... kept separate: same roundings as the per-kernel form) in
// registers: a pixel's C classes reuse them
template <int DP1>
struct CrfVerts {
    long long off[DP1];
    float w[DP1];
    __device__ __forceinline__ void load(const CrfSide& L, long long n, int C) {
#pragma unroll
        for (int j = 0; j < DP1; ++j) { off[j] = (long long)L.offset[n * DP1 + j] * C; w[j] = L.bary[n * DP1 + j]; }
    }
    __device__ __forceinline__ float slice(const float* __restrict__ lat, int k, float alpha) const {
        float v[DP1];
#pragma unroll
        for (int j = 0; j < DP1; ++j) v[j] = lat[off[j] + k];          // independent gathers, all in flight
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < DP1; ++j) s += __fmul_rn(__fmul_rn(w[j], v[j]), alpha);
        return s;
    }
};
__device__ __forceinline__ const float* crf_final_lat(const CrfSide& L) { return (L.Dp1 & 1) ? L.lat1 : L.lat0; }   // buffer the last pass wrote

// normalisers of both lattices from K 1: norm = 1 / sqrt(K 1 + 1e-20)   (DenseKernel::initLattice, NORMALIZE_SYMMETRIC); C = 1
__global__ __launch_bounds__(256) void crf_make_norm2_kernel(CrfSide g, CrfSide b, long long N, float* __restrict__ norm_g, float* __restrict__ norm_b) {
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    CrfVerts<3> vg;
    CrfVerts<6> vb;
    vg.load(g, n, 1);
    vb.load(b, n, 1);
    norm_g[n] = 1.0f / sqrtf(vg.slice(crf_final_lat(g), 0, g.alpha) + 1e-20f);
    norm_b[n] = 1.0f / sqrtf(vb.slice(crf_final_lat(b), 0, b.alpha) + 1e-20f);
}
// slice both messages + the mean-field update of one pixel:  Q = softmax(-U + w_g K_g Q + w_b K_b Q).  U [C,N] (plane-major, as
// unary_from_softmax lays it out); Q [N,C].  have_msg = 0: the initial Q = softmax(-U).
__global__ __launch_bounds__(256) void crf_update_kernel(const float* __restrict__ prob, int is_energy, long long N, int C, int have_msg, CrfSide g, float wg,
                                                         CrfSide b, float wb, float* __restrict__ Q, float* __restrict__ out_cn) {
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    CrfVerts<3> vg;
    CrfVerts<6> vb;
    float ng = 0.f, nb = 0.f;
    const float *lg = crf_final_lat(g), *lb = crf_final_lat(b);
    if (have_msg) { vg.load(g, n, C); vb.load(b, n, C); ng = g.norm[n]; nb = b.norm[n]; }
    float mx = -INFINITY;
    for (int k = 0; k < C; ++k) {
        const float pv = prob[(long long)k * N + n];
        const float u = is_energy ? pv : -logf(fminf(fmaxf(pv, 1e-5f), 1.0f));                  // unary_from_softmax (clip 1e-5)
        float t = -u;
        if (have_msg) {
            const float mg = __fmul_rn(vg.slice(lg, k, g.alpha), ng), mb = __fmul_rn(vb.slice(lb, k, b.alpha), nb);
            t = t + wg * mg + wb * mb;
        }
        Q[n * C + k] = t;
        mx = fmaxf(mx, t);
    }
    float sum = 0.f;
    for (int k = 0; k < C; ++k) { const float e = expf(Q[n * C + k] - mx); Q[n * C + k] = e; sum += e; }
    for (int k = 0; k < C; ++k) {
        const float q = Q[n * C + k] / sum;
        Q[n * C + k] = q;
        if (out_cn) out_cn[(long long)k * N + n] = q;
    }
}

struct CrfLattice {
    CrfKey *keys, *lkeys;
    float *bary, *norm;
    int *table, *rep, *latidx, *offset, *counter;
    int2* nbr;
    unsigned mask;
    long long npv;
    int D;
};
static size_t crf_al(size_t b) { return (b + 255) / 256 * 256; }
static size_t crf_lattice_bytes(long long N, int D, unsigned* cap_out) {
    const long long npv = N * (D + 1);
    unsigned cap = 1024;
    while ((long long)cap < 2 * npv) cap <<= 1;
    if (cap_out) *cap_out = cap;
    return 2 * crf_al(sizeof(CrfKey) * npv) + crf_al(4 * npv) + crf_al(4 * N) + crf_al(4ull * cap) + 3 * crf_al(4 * npv) + 256 +
           crf_al(sizeof(int2) * (size_t)npv * (D + 1));
}
static CrfLattice crf_lattice_layout(char*& p, long long N, int D) {
    CrfLattice L;
    unsigned cap;
    crf_lattice_bytes(N, D, &cap);
    const long long npv = N * (D + 1);
    auto take = [&](size_t b) { char* r = p; p += crf_al(b); return r; };
    L.keys = (CrfKey*)take(sizeof(CrfKey) * npv); L.lkeys = (CrfKey*)take(sizeof(CrfKey) * npv);
    L.bary = (float*)take(4 * npv); L.norm = (float*)take(4 * N);
    L.table = (int*)take(4ull * cap);
    L.rep = (int*)take(4 * npv); L.latidx = (int*)take(4 * npv); L.offset = (int*)take(4 * npv);
    L.counter = (int*)take(256);
    L.nbr = (int2*)take(sizeof(int2) * (size_t)npv * (D + 1));
    L.mask = cap - 1; L.npv = npv; L.D = D;
    return L;
}

extern "C" size_t excel_dcrf_workspace_bytes(int H, int W, int C) {
    const long long N = (long long)H * W;
    // per lattice: build tables + fixed-point accumulators [npv*C] + two float lattices [npv*C]; Q [N,C]; ones [N]
    return crf_lattice_bytes(N, 2, nullptr) + crf_lattice_bytes(N, 5, nullptr) + crf_al(8 * (N * 3) * C) + 2 * crf_al(4 * (N * 3) * C) +
           crf_al(8 * (N * 6) * C) + 2 * crf_al(4 * (N * 6) * C) + crf_al(4 * N * C) + crf_al(4 * N);
}

template <int D>
static int crf_build(const CrfLattice& L, const unsigned char* rgb, int H, int W, float sxy, float srgb, hipStream_t st) {
    const long long N = (long long)H * W;
    hipMemsetAsync(L.table, 0xFF, 4ull * (L.mask + 1), st);
    hipMemsetAsync(L.counter, 0, 4, st);
    hipLaunchKernelGGL(crf_lattice_kernel<D>, dim3((unsigned)cdivl(N, 256)), dim3(256), 0, st, rgb, H, W, sxy, srgb, L.keys, L.bary);
    hipLaunchKernelGGL(crf_hash_insert_kernel, dim3((unsigned)cdivl(L.npv, 256)), dim3(256), 0, st, L.keys, L.npv, L.table, L.mask, L.rep, L.latidx, L.counter, L.lkeys);
    hipLaunchKernelGGL(crf_offsets_kernel, dim3((unsigned)cdivl(L.npv, 256)), dim3(256), 0, st, L.rep, L.latidx, L.npv, L.offset);
    hipLaunchKernelGGL(crf_neighbors_kernel<D>, dim3((unsigned)cdivl(L.npv * (D + 1), 256)), dim3(256), 0, st, L.lkeys, L.counter, L.keys, L.table, L.mask,
                       L.latidx, L.nbr, L.npv);
    EXCEL_CHECK_LAUNCH("dcrf lattice build");
    return EXCEL_OK;
}

// messages of BOTH kernels for `in` [N,C]: splat, blur passes; the caller slices (crf_make_norm2 / crf_update)
static int crf_pass(const CrfSide& g, const CrfSide& b, const float* in, int use_norm, int C, hipStream_t st) {
    const long long work = (g.npv + b.npv) * C;
    const unsigned gm = (unsigned)(cdivl(work, 256) < 4096 ? cdivl(work, 256) : 4096);      // grid-stride kernels
    hipLaunchKernelGGL(crf_splat2_kernel, dim3(gm), dim3(256), 0, st, in, use_norm, g, b, C);
    const int passes = g.Dp1 > b.Dp1 ? g.Dp1 : b.Dp1;
    for (int j = 0; j < passes; ++j) hipLaunchKernelGGL(crf_blur2_kernel, dim3(gm), dim3(256), 0, st, g, b, j, C);
    EXCEL_CHECK_LAUNCH("dcrf message passing");
    return EXCEL_OK;
}

extern "C" int excel_dcrf_inference(const unsigned char* rgb_hwc, const float* prob, int prob_is_energy, int H, int W, int C, int iters, float pos_w,
                                    float pos_xy_std, float bi_w, float bi_xy_std, float bi_rgb_std, float* q_out, void* workspace, void* stream) {
    EXCEL_CHECK_ARG(rgb_hwc && prob && q_out && workspace && H > 0 && W > 0 && C >= 1 && iters >= 0, "dcrf_inference: bad argument");
    static_assert(true, "the message-passing kernels are written for the D = 2 (Gaussian) and D = 5 (bilateral) lattices of DenseCRF2D");
    EXCEL_CHECK_ARG(pos_xy_std > 0.f && bi_xy_std > 0.f && bi_rgb_std > 0.f, "dcrf_inference: standard deviations must be positive");
    hipStream_t st = (hipStream_t)stream;
    const long long N = (long long)H * W;
    char* p = (char*)workspace;
    CrfLattice Lg = crf_lattice_layout(p, N, 2), Lb = crf_lattice_layout(p, N, 5);
    CrfSide sg, sb;
    auto side = [&](CrfSide& s, const CrfLattice& L) {
        s.offset = L.offset; s.bary = L.bary; s.norm = L.norm; s.nbr = L.nbr; s.counter = L.counter; s.npv = L.npv; s.Dp1 = L.D + 1;
        s.alpha = 1.0f / (1.0f + powf(2.0f, (float)-L.D));
        s.acc = (long long*)p; p += crf_al(8 * L.npv * C);
        s.lat0 = (float*)p; p += crf_al(4 * L.npv * C);
        s.lat1 = (float*)p; p += crf_al(4 * L.npv * C);
    };
    side(sg, Lg);
    side(sb, Lb);
    float* Q = (float*)p; p += crf_al(4 * N * C);
    float* ones = (float*)p; p += crf_al(4 * N);
    TRY(crf_build<2>(Lg, rgb_hwc, H, W, pos_xy_std, 1.f, st));
    TRY(crf_build<5>(Lb, rgb_hwc, H, W, bi_xy_std, bi_rgb_std, st));
    // the accumulators are cleared once; every message pass leaves them cleared (blur pass 1)
    hipMemsetAsync(sg.acc, 0, 8ull * Lg.npv * C, st);
    hipMemsetAsync(sb.acc, 0, 8ull * Lb.npv * C, st);
    {
        const float one = 1.0f;
        unsigned pattern;
        memcpy(&pattern, &one, 4);
        hipMemsetD32Async((hipDeviceptr_t)ones, (int)pattern, (size_t)N, st);
    }
    const unsigned gn = (unsigned)cdivl(N, 256);
    // normalisers: norm = 1 / sqrt(K 1 + 1e-20)
    TRY(crf_pass(sg, sb, ones, 0, 1, st));
    hipLaunchKernelGGL(crf_make_norm2_kernel, dim3(gn), dim3(256), 0, st, sg, sb, N, Lg.norm, Lb.norm);
    hipLaunchKernelGGL(crf_update_kernel, dim3(gn), dim3(256), 0, st, prob, prob_is_energy, N, C, 0, sg, 0.f, sb, 0.f, Q, iters == 0 ? q_out : nullptr);
    for (int it = 0; it < iters; ++it) {
        TRY(crf_pass(sg, sb, Q, 1, C, st));
        hipLaunchKernelGGL(crf_update_kernel, dim3(gn), dim3(256), 0, st, prob, prob_is_energy, N, C, 1, sg, pos_w, sb, bi_w, Q, it == iters - 1 ? q_out : nullptr);
    }
    EXCEL_CHECK_LAUNCH("dcrf mean field");
    return EXCEL_OK;
}
