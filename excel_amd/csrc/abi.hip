// extern "C" entry points of libexcel_hip (declared in include/excel_hip.h) and the host-side ViT driver.
#include "../../include/excel_hip.h"
#include "common.h"
#include "excel_internal.h"

#include <stdarg.h>
#include <map>
#include <vector>

static thread_local char g_err[512] = "";

void excel_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* excel_last_error(void) { return g_err; }
extern "C" int excel_abi_version(void) { return 4; }
#ifndef EXCEL_BUILD_ID
#define EXCEL_BUILD_ID "unstamped"
#endif
extern "C" const char* excel_build_id(void) { return EXCEL_BUILD_ID; }

// ------------------------------------------------------------------------------------ profiling hooks
bool g_excel_prof_on = false;
thread_local int g_excel_prof_gemm_cat = -1;    // per host thread: concurrent forwards on different threads do not race on it
unsigned long long g_excel_prof_mask = ~0ull;
int g_excel_prof_every = 1;
unsigned g_excel_prof_seen[PROF_NCAT];
namespace {
struct ProfRec { int cat; hipEvent_t a, b; };
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
double g_prof_work[PROF_NCAT];
const char* PROF_NAMES[PROF_NCAT] = {"gemm_nt", "gemm_nn", "gemm_bf16x3", "attn_rowpass", "attn_accum", "layernorm", "embed", "token_norm",
                                     "cam_epilogue", "sinkhorn", "bbox_mask", "matvec", "cam_upsample", "par_affinity",
                                     "par_iterate", "argmax", "confusion", "other", "cam_proj", "cam_fused"};
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace
void excel_prof_begin(int cat, hipStream_t st, double work) {
    ProfRec r{cat, prof_event(), prof_event()};
    (void)hipEventRecord(r.a, st);
    g_prof_recs.push_back(r);
    g_prof_work[cat] += work;
}
void excel_prof_end(int cat, hipStream_t st) {
    for (size_t i = g_prof_recs.size(); i-- > 0;)
        if (g_prof_recs[i].cat == cat) { (void)hipEventRecord(g_prof_recs[i].b, st); return; }
}
extern "C" int excel_prof_enable(int on) {
    g_excel_prof_on = on != 0;
    return EXCEL_OK;
}
extern "C" int excel_prof_set_mask(unsigned long long mask) {
    g_excel_prof_mask = mask;
    return EXCEL_OK;
}
extern "C" int excel_prof_set_sampling(int every) {
    g_excel_prof_every = every > 0 ? every : 1;
    for (int c = 0; c < PROF_NCAT; ++c) g_excel_prof_seen[c] = 0;
    return EXCEL_OK;
}
extern "C" int excel_prof_num_categories(void) { return PROF_NCAT; }
extern "C" const char* excel_prof_category_name(int cat) { return (cat >= 0 && cat < PROF_NCAT) ? PROF_NAMES[cat] : ""; }
// Synchronises on the recorded events, sums elapsed ms / launches / algorithmic work per category, resets the log.
extern "C" int excel_prof_collect(double* ms, long long* launches, double* work) {
    for (int c = 0; c < PROF_NCAT; ++c) { ms[c] = 0.0; launches[c] = 0; work[c] = g_prof_work[c]; g_prof_work[c] = 0.0; }
    for (auto& r : g_prof_recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms[r.cat] += t;
            launches[r.cat] += 1;
        }
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    return EXCEL_OK;
}

#define ST(s) ((hipStream_t)(s))
#define TRY(x)                  \
    do {                        \
        int rc__ = (x);         \
        if (rc__) return rc__;  \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static GemmArgs gemm_args(const float* A, const float* B, float* C, const float* bias, const float* res, int M, int N, int K,
                          int lda, int ldb, int ldc, int ldr, int act) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.res = res;
    g.M = M; g.N = N; g.K = K; g.Kld = (K + 3) / 4 * 4;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
    g.act = act; g.out_mode = GEMM_OUT_PLAIN; g.alpha = 1.f; g.zdiv = 1;
    return g;
}

static GemmBfArgs gemm_bf_args(const void* A, const unsigned short* W, float* C, void* Cs, const float* bias, const float* res,
                               int M, int N, int K, int ldc, int ldr, int act, int out_mode) {
    GemmBfArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const unsigned short*)A; g.B = W; g.C = C; g.Cs = (unsigned short*)Cs; g.bias = bias; g.res = res;
    g.M = M; g.N = N; g.K = K; g.lda = 2 * K; g.ldb = 2 * K; g.ldc = ldc; g.ldr = ldr; g.act = act; g.out_mode = out_mode;
    return g;
}

// ------------------------------------------------------------------------------------ small helper kernels
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < Cc) t[k][tx] = in[(long long)(r0 + k) * Cc + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < Cc && r0 + tx < R) out[(long long)(c0 + k) * R + r0 + tx] = t[tx][k];
}

// positional grid resize, F.interpolate(bilinear, align_corners=False) on [1,D,side,side] (clip_surgery_model.py:430-433)
__global__ void pos_resize_kernel(const float* __restrict__ pos, float* __restrict__ out, int side, int g, int D) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)(g * g + 1) * D;
    if (i >= total) return;
    const int d = (int)(i % D);
    const int n = (int)(i / D);
    if (n == 0) { out[i] = pos[d]; return; }
    const int y = (n - 1) / g, x = (n - 1) % g;
    const float sc = (float)side / (float)g;
    float fy = fmaxf(sc * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sc * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)fy, side - 1), x0 = min((int)fx, side - 1);
    const int y1 = min(y0 + 1, side - 1), x1 = min(x0 + 1, side - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    auto at = [&](int yy, int xx) { return pos[(long long)(1 + yy * side + xx) * D + d]; };
    out[i] = (1.f - ly) * ((1.f - lx) * at(y0, x0) + lx * at(y0, x1)) + ly * ((1.f - lx) * at(y1, x0) + lx * at(y1, x1));
}

__global__ void attn_layer_mean_kernel(const float* __restrict__ attn, int B, int N, int first, int nl, float* __restrict__ out) {
    const long long P = N - 1;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * P * P) return;
    const int c = (int)(i % P), r = (int)((i / P) % P);
    const long long b = i / (P * P);
    float s = 0.f;
    for (int l = 0; l < nl; ++l) s += attn[(((long long)(first + l) * B + b) * N + (r + 1)) * N + (c + 1)];
    out[i] = s / (float)nl;
}

// ------------------------------------------------------------------------------------ building blocks
extern "C" int excel_gemm_f32(const float* A, const float* Bm, float* C, const float* bias, const float* residual, int M, int N,
                              int K, int lda, int ldb, int ldc, int ldr, int b_kmajor, int act, int batch, long long sA,
                              long long sB, long long sC, long long sR, void* stream) {
    GemmArgs g = gemm_args(A, Bm, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, act);
    EXCEL_CHECK_ARG((K % 4) == 0 || !b_kmajor, "excel_gemm_f32: K must be a multiple of 4 in NT mode (K=%d)", K);
    if (!b_kmajor) {
        EXCEL_CHECK_ARG((K % 4) == 0, "excel_gemm_f32: K must be a multiple of 4 (pad A with zeros) (K=%d)", K);
    }
    g.sA = sA; g.sB = sB; g.sC = sC; g.sR = sR;
    return excel_launch_gemm(g, b_kmajor != 0, batch, ST(stream));
}

extern "C" int excel_split_bf16(const float* in, void* out, long long rows, int K, void* stream) {
    EXCEL_CHECK_ARG(in && out && rows > 0 && K > 0, "split_bf16: bad argument");
    return excel_launch_split_bf16(in, out, rows, K, ST(stream));
}

extern "C" int excel_gemm_bf16x3(const void* A_split, const void* W_split, float* C, const float* bias, const float* residual,
                                 int M, int N, int K, int act, int split_out, void* stream) {
    EXCEL_CHECK_ARG(A_split && W_split && C, "gemm_bf16x3: null argument");
    GemmBfArgs g = gemm_bf_args(A_split, (const unsigned short*)W_split, C, C, bias, residual, M, N, K, N, N, act,
                                split_out ? GEMM_OUT_SPLIT_BF16 : GEMM_OUT_PLAIN);
    return excel_launch_gemm_bf16x3(g, ST(stream));
}

extern "C" int excel_split_f16(const float* in, void* out, long long rows, int K, void* stream) {
    EXCEL_CHECK_ARG(in && out && rows > 0 && K > 0, "split_f16: bad argument");
    return excel_f16::excel_launch_split_bf16(in, out, rows, K, ST(stream));
}

extern "C" int excel_gemm_f16x3(const void* A_split, const void* W_split, float* C, const float* bias, const float* residual,
                                int M, int N, int K, int act, int split_out, void* stream) {
    EXCEL_CHECK_ARG(A_split && W_split && C, "gemm_f16x3: null argument");
    GemmBfArgs g = gemm_bf_args(A_split, (const unsigned short*)W_split, C, C, bias, residual, M, N, K, N, N, act,
                                split_out ? GEMM_OUT_SPLIT_BF16 : GEMM_OUT_PLAIN);
    return excel_f16::excel_launch_gemm_bf16x3(g, ST(stream));
}

extern "C" int excel_pack_f16(const float* in, void* out, long long rows, int K, unsigned long long* inexact_dev, void* stream) {
    EXCEL_CHECK_ARG(in && out && inexact_dev && rows > 0 && K > 0, "pack_f16: bad argument");
    return excel_f16::excel_launch_pack_hi(in, out, rows, K, inexact_dev, ST(stream));
}

extern "C" int excel_gemm_f16x2(const void* A_split, const void* W_split, const void* W_half, float* C, const float* bias,
                                const float* residual, int M, int N, int K, int act, int split_out, void* stream) {
    EXCEL_CHECK_ARG(A_split && W_split && C, "gemm_f16x2: null argument");
    GemmBfArgs g = gemm_bf_args(A_split, (const unsigned short*)W_split, C, C, bias, residual, M, N, K, N, N, act,
                                split_out ? GEMM_OUT_SPLIT_BF16 : GEMM_OUT_PLAIN);
    g.w_lo_zero = 1;
    g.Bh = (const unsigned short*)W_half;
    g.ldbh = K;
    return excel_f16::excel_launch_gemm_bf16x3(g, ST(stream));
}

extern "C" int excel_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int D, float eps, void* stream) {
    return excel_launch_layernorm(x, nullptr, 1, w, b, y, rows, D, eps, ST(stream));
}

// ------------------------------------------------------------------------------------ ViT handle
struct SplitBlockW {
    unsigned short *in_proj, *out_proj, *fc1, *fc2;             // split planes [N][2][K]
    unsigned short *h_in_proj, *h_out_proj, *h_fc1, *h_fc2;     // plain half matrices [N][K] (mode 3, "f16x2"), else null
};
struct excel_vit {
    excel_vit_config cfg;
    excel_vit_weights w;
    std::vector<excel_vit_block_weights> blocks;
    float* projT = nullptr;             // [C, D]
    std::map<int, float*> pos_cache;    // g -> [1+g*g, D]
    int gemm_mode = 0;                  // 0: exact fp32 MFMA, 1: bf16x3 (split bf16, 3 MFMAs per product), 2: f16x3 (split IEEE half), 3: f16x2 (2 + fp16-valued weights)
    int split_type = 0;                 // what the split weights hold: 0 nothing yet, 1 bf16 planes, 2 f16 planes
    unsigned short* split_arena = nullptr;   // all split weights in one allocation
    std::vector<SplitBlockW> sblocks;
    unsigned short *s_conv1 = nullptr, *s_projT = nullptr;
    // "f16x2": the weights as plain half matrices (one allocation) and whether they are all fp16-valued (-1: not examined yet)
    unsigned short* half_arena = nullptr;
    unsigned short *h_conv1 = nullptr, *h_projT = nullptr;
    int fp16_exact = -1;
};

static int vit_prepare_split_weights(excel_vit* h, int type) {
    if (h->split_arena && h->split_type == type) return EXCEL_OK;
    const excel_vit_config& c = h->cfg;
    const size_t D = c.width, Kc = (size_t)3 * c.patch * c.patch;
    const size_t per_block = 3 * D * D + D * D + 4 * D * D + 4 * D * D;       // floats == split bytes / 4
    const size_t total = per_block * c.layers + D * Kc + (size_t)c.out_dim * D;
    float* arena = (float*)h->split_arena;
    if (!arena) {
        if (hipMalloc(&arena, total * sizeof(float)) != hipSuccess) {
            excel_set_error("excel_vit: hipMalloc(split weights) failed");
            return EXCEL_ERR_ALLOC;
        }
    } else if (hipDeviceSynchronize() != hipSuccess) {      // re-splitting for the other type: nothing may still read the old planes
        excel_set_error("excel_vit: device synchronize failed");
        return EXCEL_ERR_LAUNCH;
    }
    h->split_arena = (unsigned short*)arena;
    h->split_type = 0;               // the arena holds nothing usable until every split kernel has finished (set below, after the sync)
    float* cur = arena;
    int put_rc = EXCEL_OK;
    auto put = [&](const float* src, size_t rows, size_t K) -> unsigned short* {
        unsigned short* dst = (unsigned short*)cur;
        const int rc = type == 2 ? excel_f16::excel_launch_split_bf16(src, dst, (long long)rows, (int)K, 0)
                                 : excel_bf16::excel_launch_split_bf16(src, dst, (long long)rows, (int)K, 0);
        if (rc != EXCEL_OK && put_rc == EXCEL_OK) put_rc = rc;
        cur += rows * K;
        return dst;
    };
    if (h->sblocks.size() != (size_t)c.layers) h->sblocks.assign(c.layers, SplitBlockW{});
    for (int l = 0; l < c.layers; ++l) {
        const excel_vit_block_weights& bw = h->blocks[l];
        h->sblocks[l].in_proj = put(bw.in_proj_w, 3 * D, D);
        h->sblocks[l].out_proj = put(bw.out_proj_w, D, D);
        h->sblocks[l].fc1 = put(bw.fc1_w, 4 * D, D);
        h->sblocks[l].fc2 = put(bw.fc2_w, D, 4 * D);
    }
    h->s_conv1 = put(h->w.conv1_w, D, Kc);
    h->s_projT = put(h->projT, c.out_dim, D);
    // (one-time set-up on the legacy stream 0, which every blocking stream orders against; the device-wide sync above / this one make it
    // safe for callers on non-blocking streams too)
    if (put_rc != EXCEL_OK) return put_rc;
    if (hipStreamSynchronize(0) != hipSuccess) {
        excel_set_error("excel_vit: splitting weights failed: %s", hipGetErrorString(hipGetLastError()));
        return EXCEL_ERR_LAUNCH;
    }
    h->split_type = type;            // only now: a failed re-split leaves split_type 0, so the next set_gemm_mode splits again
    return EXCEL_OK;
}

// Packs every GEMM weight as a plain half matrix (the compact operand of the two-product GEMM) and counts the elements that are not
// exactly representable: fp16_exact = (count == 0).  One pass, once per handle.
static int vit_prepare_half_weights(excel_vit* h) {
    if (h->fp16_exact >= 0) return EXCEL_OK;
    const excel_vit_config& c = h->cfg;
    const size_t D = c.width, Kc = (size_t)3 * c.patch * c.patch;
    const size_t per_block = 3 * D * D + D * D + 4 * D * D + 4 * D * D;
    const size_t total = per_block * c.layers + D * Kc + (size_t)c.out_dim * D;        // halfs
    unsigned short* arena = nullptr;
    unsigned long long* cnt = nullptr;
    if (hipMalloc(&arena, total * sizeof(unsigned short) + 256) != hipSuccess) {
        excel_set_error("excel_vit: hipMalloc(half weights) failed");
        return EXCEL_ERR_ALLOC;
    }
    cnt = (unsigned long long*)((char*)arena + align_up(total * sizeof(unsigned short), 16));
    if (hipMemsetAsync(cnt, 0, sizeof(unsigned long long), 0) != hipSuccess) { hipFree(arena); excel_set_error("excel_vit: memset failed"); return EXCEL_ERR_LAUNCH; }
    unsigned short* cur = arena;
    int put_rc = EXCEL_OK;
    auto put = [&](const float* src, size_t rows, size_t K) -> unsigned short* {
        unsigned short* dst = cur;
        const int rc = excel_f16::excel_launch_pack_hi(src, dst, (long long)rows, (int)K, cnt, 0);
        if (rc != EXCEL_OK && put_rc == EXCEL_OK) put_rc = rc;
        cur += rows * K;
        return dst;
    };
    if (h->sblocks.size() != (size_t)c.layers) h->sblocks.assign(c.layers, SplitBlockW{});
    for (int l = 0; l < c.layers; ++l) {
        const excel_vit_block_weights& bw = h->blocks[l];
        h->sblocks[l].h_in_proj = put(bw.in_proj_w, 3 * D, D);
        h->sblocks[l].h_out_proj = put(bw.out_proj_w, D, D);
        h->sblocks[l].h_fc1 = put(bw.fc1_w, 4 * D, D);
        h->sblocks[l].h_fc2 = put(bw.fc2_w, D, 4 * D);
    }
    h->h_conv1 = put(h->w.conv1_w, D, Kc);
    h->h_projT = put(h->projT, c.out_dim, D);
    unsigned long long n_bad = 1;
    if (put_rc != EXCEL_OK || hipMemcpy(&n_bad, cnt, sizeof(n_bad), hipMemcpyDeviceToHost) != hipSuccess) {       // (synchronises)
        for (auto& sb : h->sblocks) sb.h_in_proj = sb.h_out_proj = sb.h_fc1 = sb.h_fc2 = nullptr;
        h->h_conv1 = h->h_projT = nullptr;
        hipFree(arena);
        if (put_rc != EXCEL_OK) return put_rc;
        excel_set_error("excel_vit: packing half weights failed: %s", hipGetErrorString(hipGetLastError()));
        return EXCEL_ERR_LAUNCH;
    }
    h->fp16_exact = n_bad == 0 ? 1 : 0;
    if (h->fp16_exact) {
        h->half_arena = arena;
    } else {                           // full-mantissa weights: the packed matrices are of no use (the two-product mode is refused)
        for (auto& sb : h->sblocks) sb.h_in_proj = sb.h_out_proj = sb.h_fc1 = sb.h_fc2 = nullptr;
        h->h_conv1 = h->h_projT = nullptr;
        hipFree(arena);
    }
    return EXCEL_OK;
}

extern "C" int excel_vit_weights_fp16_exact(excel_vit_t h) {
    EXCEL_CHECK_ARG(h, "excel_vit_weights_fp16_exact: null handle");
    if ((h->cfg.width % 32) != 0 || ((3 * h->cfg.patch * h->cfg.patch) % 32) != 0) return 0;      // no split-plane mode for this shape at all
    TRY(vit_prepare_half_weights(h));
    return h->fp16_exact;
}

extern "C" int excel_vit_set_gemm_mode(excel_vit_t h, int mode) {
    EXCEL_CHECK_ARG(h && mode >= 0 && mode <= 3, "excel_vit_set_gemm_mode: mode must be 0 (f32), 1 (bf16x3), 2 (f16x3) or 3 (f16x2)");
    if (mode >= 1) {
        EXCEL_CHECK_ARG((h->cfg.width % 32) == 0 && ((3 * h->cfg.patch * h->cfg.patch) % 32) == 0,
                        "the split-plane modes need width and 3*patch^2 to be multiples of 32");
        if (mode == 3) {
            TRY(vit_prepare_half_weights(h));
            EXCEL_CHECK_ARG(h->fp16_exact == 1, "excel_vit_set_gemm_mode: mode 3 (f16x2) needs fp16-valued weights (excel_vit_weights_fp16_exact); use 2 (f16x3)");
        }
        TRY(vit_prepare_split_weights(h, mode == 3 ? 2 : mode));
    }
    h->gemm_mode = mode;
    return EXCEL_OK;
}
extern "C" int excel_vit_get_gemm_mode(excel_vit_t h) { return h ? h->gemm_mode : -1; }



extern "C" int excel_vit_create(const excel_vit_config* cfg, const excel_vit_weights* w, excel_vit_t* out) {
    EXCEL_CHECK_ARG(cfg && w && out, "excel_vit_create: null argument");
    EXCEL_CHECK_ARG(cfg->heads > 0 && cfg->width == cfg->heads * 64, "excel_vit_create: head_dim must be 64 (width=%d heads=%d)", cfg->width, cfg->heads);
    EXCEL_CHECK_ARG(cfg->layers >= 1 && cfg->n_surgery >= 0 && cfg->n_surgery <= cfg->layers, "excel_vit_create: bad layer counts");
    EXCEL_CHECK_ARG((cfg->out_dim % 4) == 0 && (cfg->patch % 4) == 0, "excel_vit_create: out_dim and patch must be multiples of 4");
    excel_vit* h = new excel_vit();
    h->cfg = *cfg;
    h->w = *w;
    h->blocks.assign(w->blocks, w->blocks + cfg->layers);
    h->w.blocks = h->blocks.data();
    if (hipMalloc(&h->projT, sizeof(float) * cfg->out_dim * cfg->width) != hipSuccess) {
        delete h;
        excel_set_error("excel_vit_create: hipMalloc failed");
        return EXCEL_ERR_ALLOC;
    }
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cfg->out_dim, 32), cdiv(cfg->width, 32)), dim3(256), 0, 0, w->proj, h->projT,
                       cfg->width, cfg->out_dim);
    if (hipStreamSynchronize(0) != hipSuccess) {
        excel_set_error("excel_vit_create: transpose failed: %s", hipGetErrorString(hipGetLastError()));
        hipFree(h->projT);
        delete h;
        return EXCEL_ERR_LAUNCH;
    }
    *out = h;
    const char* mode = getenv("EXCEL_GEMM_MODE");     // default numerics of new handles: "f32" | "bf16x3" | "f16x3" | "f16x2"
    if (mode && (!strcmp(mode, "bf16x3") || !strcmp(mode, "f16x3") || !strcmp(mode, "f16x2")) && (cfg->width % 32) == 0 && ((3 * cfg->patch * cfg->patch) % 32) == 0) {
        int rc = excel_vit_set_gemm_mode(h, !strcmp(mode, "f16x2") ? 3 : !strcmp(mode, "f16x3") ? 2 : 1);
        if (rc) return rc;
    }
    return EXCEL_OK;
}

extern "C" void excel_vit_destroy(excel_vit_t h) {
    if (!h) return;
    if (h->projT) hipFree(h->projT);
    if (h->split_arena) hipFree(h->split_arena);
    if (h->half_arena) hipFree(h->half_arena);
    for (auto& kv : h->pos_cache) hipFree(kv.second);
    delete h;
}

struct VitWs {
    float *x, *xo, *y, *ao, *qkvh, *qkvs, *hbuf, *stats, *a_sum, *vt, *fraw, *ss;
    int NP, KP;
    size_t total;
};

static VitWs vit_ws_layout(const excel_vit_config& c, int B, int S, char* base) {
    const int g = S / c.patch, N = g * g + 1;
    const size_t M = (size_t)B * N, D = c.width;
    VitWs w;
    w.NP = (N + 3) / 4 * 4;
    w.KP = (N + 31) / 32 * 32;            // K of the bf16x3 A_sum.V GEMM (zero padded)
    size_t off = 0;
    auto take = [&](size_t floats) { float* p = (float*)(base + off); off += align_up(floats * sizeof(float), 256); return p; };
    w.x = take(M * D);
    w.xo = take(M * D);
    w.y = take(M * D);
    w.ao = take(M * D);                   // also the patch-embed GEMM output [B*P, D]
    w.qkvh = take(M * 3 * D);
    w.qkvs = take(M * 3 * D);             // split-bf16 copy of q|k|v for the bf16x3 attention scores (bf16x3 mode)
    {   // MLP hidden [M,4D]; also holds the im2col matrix [B*P, 3*ps*ps]
        const size_t col = (size_t)B * (N - 1) * 3 * c.patch * c.patch;
        w.hbuf = take(M * 4 * D > col ? M * 4 * D : col);
    }
    w.stats = take((size_t)B * c.heads * 4 * N * 2);
    w.a_sum = take((size_t)B * N * w.KP);              // fp32 [B,N,NP] or split bf16 [B,N][2*KP]
    w.vt = take((size_t)B * D * w.KP);                 // V^T split [B][D][2*KP] (bf16x3 mode)
    w.fraw = take(M * c.out_dim);
    w.ss = take((size_t)B * c.out_dim);
    w.total = off;
    return w;
}

extern "C" size_t excel_vit_workspace_bytes(excel_vit_t h, int B, int S) {
    if (!h || B <= 0 || S <= 0 || S % h->cfg.patch) return 0;
    return vit_ws_layout(h->cfg, B, S, nullptr).total;
}

static int vit_forward_impl(excel_vit_t h, const float* img, int B, int S, void* workspace, size_t workspace_bytes,
                            float* image_features, float* x_raw, float* w_aff, int aff_layers, float* attn_out,
                            int n_attn_out, float* feats_out, const float* ex_attn, int flags, void* stream);

extern "C" int excel_vit_forward(excel_vit_t h, const float* img, int B, int S, void* workspace, size_t workspace_bytes,
                                 float* image_features, float* x_raw, float* w_aff, int aff_layers, float* attn_out,
                                 int n_attn_out, float* feats_out, void* stream) {
    return vit_forward_impl(h, img, B, S, workspace, workspace_bytes, image_features, x_raw, w_aff, aff_layers, attn_out, n_attn_out,
                            feats_out, nullptr, 0, stream);
}

extern "C" int excel_vit_forward_ex(excel_vit_t h, const float* img, int B, int S, void* workspace, size_t workspace_bytes,
                                    float* image_features, float* x_raw, float* w_aff, int aff_layers, float* attn_out,
                                    int n_attn_out, float* feats_out, const float* ex_attn, int flags, void* stream) {
    return vit_forward_impl(h, img, B, S, workspace, workspace_bytes, image_features, x_raw, w_aff, aff_layers, attn_out, n_attn_out,
                            feats_out, ex_attn, flags, stream);
}

extern "C" size_t excel_train_losses_workspace_bytes(int B, int nc, int H, int W) { return excel_train_losses_ws_bytes(B, nc, H, W); }

extern "C" int excel_train_losses(const float* seg, const float* attn_pred, const unsigned char* pseudo, const unsigned char* aff_labels, int B, int nc,
                                  int g_h, int g_w, int H, int W, int radius, int ignore_index, float w_seg, float w_diver, float* losses,
                                  float* d_seg, float* d_attn_pred, void* workspace, void* stream) {
    return excel_launch_train_losses(seg, attn_pred, pseudo, aff_labels, B, nc, g_h, g_w, H, W, radius, ignore_index, w_seg, w_diver, losses, d_seg,
                                     d_attn_pred, workspace, ST(stream));
}

extern "C" int excel_lam_to_label(const float* cam, const float* cls_label, const int32_t* img_box, int B, int F, int H, int W, float bkg_thre,
                                 float high_thre, float low_thre, int ignore_mid, int ignore_index, float* valid_cam, unsigned char* label,
                                 void* stream) {
    EXCEL_CHECK_ARG(cam && cls_label && label && B > 0 && F > 0 && H > 0 && W > 0, "lam_to_label: bad argument");
    return excel_launch_lam_to_label(cam, cls_label, img_box, B, F, H, W, bkg_thre, high_thre, low_thre, ignore_mid, ignore_index, valid_cam, label,
                                     ST(stream));
}

extern "C" int excel_normalize_img_u8(const unsigned char* hwc, int B, int H, int W, const double* mean3, const double* std3, float* out, void* stream) {
    EXCEL_CHECK_ARG(hwc && out && mean3 && std3 && B > 0 && H > 0 && W > 0, "normalize_img_u8: bad argument");
    return excel_launch_normalize_u8(hwc, out, B, (long long)H * W, mean3, std3, ST(stream));
}

extern "C" int excel_denormalize_img(const float* img, int B, int H, int W, const float* mean3, const float* std3, unsigned char* out_u8,
                                    float* out_f32, void* stream) {
    EXCEL_CHECK_ARG(img && mean3 && std3 && (out_u8 || out_f32) && B > 0 && H > 0 && W > 0, "denormalize_img: bad argument");
    return excel_launch_denormalize(img, out_u8, out_f32, B, (long long)H * W, mean3, std3, ST(stream));
}

extern "C" int excel_seg_scale_accumulate(const float* segs, float* acc, int B, int nc, int h, int w, int H, int W, int flip_mean,
                                         int init, float scale, void* stream) {
    EXCEL_CHECK_ARG(segs && acc && B > 0 && nc > 0 && h > 0 && w > 0 && H > 0 && W > 0, "seg_scale_accumulate: bad argument");
    return excel_launch_seg_scale_accumulate(segs, acc, B, nc, h, w, H, W, flip_mean, init, scale, ST(stream));
}

extern "C" size_t excel_feature_affinity_workspace_bytes(int B, int C, int P) { return excel_feature_affinity_ws_bytes(B, C, P); }

extern "C" int excel_feature_affinity(const float* feats, int B, int C, int P, float beta, float gamma, int mode, float* out,
                                      void* workspace, void* stream) {
    return excel_launch_feature_affinity(feats, B, C, P, beta, gamma, mode, out, workspace, ST(stream));
}

extern "C" size_t excel_attn_select_workspace_bytes(int B, int n_layers) { return excel_attn_select_ws_bytes(B, n_layers); }

extern "C" int excel_attn_select_mean(const float* attn, int Lw, int B, int N, int first_layer, int n_layers, const float* seg_attn,
                                      float* w_out, void* workspace, void* stream) {
    return excel_launch_attn_select_mean(attn, Lw, B, N, first_layer, n_layers, seg_attn, w_out, workspace, ST(stream));
}

#define EXCEL_VIT_FWD_ARGS excel_vit_t h, const float* img, int B, int S, void* workspace, size_t workspace_bytes, float* image_features, \
    float* x_raw, float* w_aff, int aff_layers, float* attn_out, int n_attn_out, float* feats_out, const float* ex_attn, int flags, void* stream
static int vit_forward_bf16(EXCEL_VIT_FWD_ARGS) {
#include "vit_forward_body.inc"
}
static int vit_forward_f16(EXCEL_VIT_FWD_ARGS) {
    // IEEE-half split planes ("f16x3"): every split-type dependent launcher of the body comes from namespace excel_f16
    using excel_f16::excel_launch_gemm; using excel_f16::excel_launch_gemm_bf16x3; using excel_f16::excel_launch_split_bf16;
    using excel_f16::excel_launch_vt_from_planes; using excel_f16::excel_launch_layernorm; using excel_f16::excel_launch_assemble_ln_pre;
    using excel_f16::excel_launch_token_axis_normalize; using excel_f16::excel_launch_im2col; using excel_f16::excel_launch_attn_rowpass;
    using excel_f16::excel_launch_attn_accum; using excel_f16::excel_attn_strip_supported; using excel_f16::excel_launch_attn_strip;
#include "vit_forward_body.inc"
}
static int vit_forward_impl(EXCEL_VIT_FWD_ARGS) {
    EXCEL_CHECK_ARG(h, "excel_vit_forward: null handle");
    return h->gemm_mode >= 2 ? vit_forward_f16(h, img, B, S, workspace, workspace_bytes, image_features, x_raw, w_aff, aff_layers, attn_out,
                                               n_attn_out, feats_out, ex_attn, flags, stream)
                             : vit_forward_bf16(h, img, B, S, workspace, workspace_bytes, image_features, x_raw, w_aff, aff_layers, attn_out,
                                                n_attn_out, feats_out, ex_attn, flags, stream);
}

// ------------------------------------------------------------------------------------ CAM
extern "C" size_t excel_cam_workspace_bytes(int B, int N, int T) {
    return align_up((size_t)B * N * ((T + 3) / 4 * 4) * sizeof(float), 256);
}

extern "C" int excel_clip_feature_surgery(const float* image_features, const float* text, int B, int N, int C, int T, int F,
                                          float temperature, float* out_full, float* out_slice, void* workspace, void* stream) {
    EXCEL_CHECK_ARG(image_features && text && workspace && (out_full || out_slice), "clip_feature_surgery: null argument");
    EXCEL_CHECK_ARG((C % 4) == 0, "clip_feature_surgery: C must be a multiple of 4");
    const int ldT = (T + 3) / 4 * 4;
    float* S = (float*)workspace;
    // S[b*N+n, t] = f . text[t]  -- one NT GEMM over all B*N token rows (text shared)
    GemmArgs ga = gemm_args(image_features, text, S, nullptr, nullptr, B * N, T, C, C, C, ldT, 0, GEMM_ACT_NONE);
    TRY(excel_launch_gemm(ga, true, 1, ST(stream)));
    return excel_launch_cam_epilogue(S, out_full, out_slice, B, N, T, ldT, F, temperature, ST(stream));
}

// Fused path (cam.hip): token-axis norm + similarity on the matrix core + surgery epilogue, straight from the un-normalised token
// features excel_vit_forward returns as x_raw (+ their token-axis sums of squares x_colsq when the forward produced them).
struct PtcWs { float *sim, *part, *colsq; unsigned short* ts; size_t total; };
static PtcWs ptc_ws_layout(int B, int N, int C, int T, char* base) {
    PtcWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base + off; off += align_up(bytes, 256); return p; };
    w.sim = (float*)take(excel_patch_text_cam_ws_floats(B, N, C, T, 0) * sizeof(float));
    w.ts = (unsigned short*)take((size_t)T * C * sizeof(float));
    w.part = (float*)take(excel_patch_text_cam_ws_floats(B, N, C, T, 1) * sizeof(float));
    w.colsq = (float*)take(excel_patch_text_cam_ws_floats(B, N, C, T, 2) * sizeof(float));
    w.total = off;
    return w;
}
extern "C" size_t excel_patch_text_cam_workspace_bytes(int B, int N, int C, int T) { return ptc_ws_layout(B, N, C, T, nullptr).total; }

extern "C" int excel_patch_text_cam(const float* x_raw, const float* text, int B, int N, int C, int T, int F, float temperature, int mode,
                                    float* out_full, float* out_slice, float* image_features, void* workspace, void* stream) {
    EXCEL_CHECK_ARG(x_raw && text && workspace && (out_full || out_slice), "patch_text_cam: null argument");
    EXCEL_CHECK_ARG(mode == 0 || mode == 1 || mode == 2, "patch_text_cam: mode must be 0 (exact fp32), 1 (bf16x3) or 2 (f16x3)");
    EXCEL_CHECK_ARG(B > 0 && N > 0 && C > 0 && T > 0, "patch_text_cam: bad shape");
    const int ldT = (T + 3) / 4 * 4;
    const PtcWs w = ptc_ws_layout(B, N, C, T, (char*)workspace);
    if (mode == 2)
        return excel_f16::excel_launch_patch_text_cam(x_raw, text, w.ts, w.sim, w.part, w.colsq, out_full, out_slice, image_features, B, N, C, T, F,
                                                      ldT, temperature, 1, ST(stream));
    return excel_launch_patch_text_cam(x_raw, text, mode == 1 ? w.ts : nullptr, w.sim, w.part, w.colsq, out_full, out_slice, image_features, B, N, C,
                                       T, F, ldT, temperature, mode, ST(stream));
}

// ------------------------------------------------------------------------------------ affinity
extern "C" int excel_attn_layer_mean(const float* attn, int Lw, int B, int N, int first_layer, int n_layers, float* w_aff, void* stream) {
    EXCEL_CHECK_ARG(attn && w_aff && first_layer >= 0 && n_layers >= 1 && first_layer + n_layers <= Lw, "attn_layer_mean: bad layer range");
    const long long n = (long long)B * (N - 1) * (N - 1);
    hipLaunchKernelGGL(attn_layer_mean_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, ST(stream), attn, B, N, first_layer, n_layers, w_aff);
    EXCEL_CHECK_LAUNCH("attn_layer_mean");
    return EXCEL_OK;
}

extern "C" size_t excel_trans_mat_workspace_bytes(int B, int P) {
    return 2 * align_up((size_t)B * P * P * sizeof(float), 256) + align_up((size_t)B * P * sizeof(float), 256);
}

extern "C" int excel_compute_trans_mat(const float* w_aff, int B, int P, float* trans_out, void* workspace, void* stream) {
    EXCEL_CHECK_ARG(w_aff && trans_out && workspace, "compute_trans_mat: null argument");
    EXCEL_CHECK_ARG((P % 4) == 0, "compute_trans_mat: P must be a multiple of 4 (P=%d)", P);
    char* base = (char*)workspace;
    const size_t mat = align_up((size_t)B * P * P * sizeof(float), 256);
    float* T = (float*)base;
    float* Tsym = (float*)(base + mat);
    float* cs = (float*)(base + 2 * mat);
    TRY(excel_launch_trans_mat_sym(w_aff, T, Tsym, cs, B, P, ST(stream)));
    // Tsym . Tsym ; Tsym symmetric => B operand [N,K] = Tsym itself (NT form)
    GemmArgs ga = gemm_args(Tsym, Tsym, trans_out, nullptr, nullptr, P, P, P, P, P, P, 0, GEMM_ACT_NONE);
    ga.sA = ga.sB = ga.sC = (long long)P * P;
    return excel_launch_gemm(ga, true, B, ST(stream));
}

extern "C" int excel_cls_compact(const float* onehot, int B, int F, int Smax, int32_t* cls_idx, int32_t* ncls, int32_t* nchan,
                                 void* stream) {
    EXCEL_CHECK_ARG(onehot && cls_idx && ncls && Smax >= 1, "cls_compact: bad argument");
    return excel_launch_cls_compact(onehot, B, F, Smax, cls_idx, ncls, nchan, ST(stream));
}

extern "C" int excel_scoremap_box_mask(const float* attr, const int32_t* cls_idx, const int32_t* ncls, int B, int g, int F, int Smax,
                                       double caa_thre, float* v_out, uint8_t* mask_out, void* stream) {
    EXCEL_CHECK_ARG(attr && cls_idx && ncls && v_out, "scoremap_box_mask: null argument");
    return excel_launch_bbox_mask(attr, cls_idx, ncls, B, g, F, Smax, caa_thre, v_out, mask_out, ST(stream));
}

extern "C" size_t excel_refine_workspace_bytes(int B, int P, int Smax) {
    return 2 * align_up((size_t)B * P * P * sizeof(float), 256) + align_up((size_t)B * P * sizeof(float), 256) +
           2 * align_up((size_t)B * Smax * P * sizeof(float), 256);
}

extern "C" int excel_refine_cams_with_aff(const float* attr, const float* w_aff, const int32_t* cls_idx, const int32_t* ncls, int B,
                                          int g, int F, int Smax, double caa_thre, float* refined, void* workspace, void* stream) {
    EXCEL_CHECK_ARG(attr && w_aff && cls_idx && ncls && refined && workspace, "refine_cams_with_aff: null argument");
    const int P = g * g;
    char* base = (char*)workspace;
    const size_t mat = align_up((size_t)B * P * P * sizeof(float), 256);
    const size_t vec = align_up((size_t)B * Smax * P * sizeof(float), 256);
    float* T = (float*)base;
    float* Tsym = (float*)(base + mat);
    float* cs = (float*)(base + 2 * mat);
    float* v = (float*)(base + 2 * mat + align_up((size_t)B * P * sizeof(float), 256));
    float* u = (float*)((char*)v + vec);
    hipStream_t st = ST(stream);
    TRY(excel_launch_trans_mat_sym(w_aff, T, Tsym, cs, B, P, st));
    TRY(excel_launch_bbox_mask(attr, cls_idx, ncls, B, g, F, Smax, caa_thre, v, nullptr, st));
    TRY(excel_launch_matvec(Tsym, v, ncls, u, B, P, Smax, st));          // u = Tsym (mask.g)
    TRY(excel_launch_matvec(Tsym, u, ncls, refined, B, P, Smax, st));    // refined = Tsym u = (Tsym.Tsym)(mask.g)
    return EXCEL_OK;
}

extern "C" int excel_cam_upsample_bkg(const float* refined, const int32_t* ncls, int B, int g, int Smax, int H, int W, float* cams,
                                      void* workspace, int flags, void* stream) {
    EXCEL_CHECK_ARG(refined && ncls && cams && workspace, "cam_upsample_bkg: null argument");
    return excel_launch_cam_upsample_bkg(refined, ncls, (float*)workspace, cams, B, g, Smax, H, W, (flags & EXCEL_CAMS_ZERO_UNUSED) ? 1 : 0, ST(stream));
}

// ------------------------------------------------------------------------------------ ragged batches (images of different sizes)
extern "C" int excel_ragged_plan(const int32_t* hw, int B, excel_ragged_info* info, int32_t* table) {
    EXCEL_CHECK_ARG(hw && info && B >= 1, "ragged_plan: bad argument");
    long long pix = 0, lab = 0, tiles = 0, max_plane = 0;
    for (int b = 0; b < B; ++b) {
        const long long H = hw[2 * b], W = hw[2 * b + 1];
        EXCEL_CHECK_ARG(H >= 1 && W >= 1, "ragged_plan: image %d has size %lld x %lld", b, H, W);
        const long long Wp = (W + 3) / 4 * 4, plane = H * Wp, nt = ((W + 63) / 64) * ((H + 15) / 16);
        if (table) {
            int32_t* rec = table + EXCEL_RAG_REC * b;
            rec[0] = (int32_t)H; rec[1] = (int32_t)W; rec[2] = (int32_t)pix; rec[3] = (int32_t)tiles; rec[4] = (int32_t)lab; rec[5] = rec[6] = rec[7] = 0;
            for (long long t = 0; t < nt; ++t) table[EXCEL_RAG_REC * (B + 1) + tiles + t] = b;
        }
        pix += plane; lab += H * W; tiles += nt;
        if (plane > max_plane) max_plane = plane;
        EXCEL_CHECK_ARG(pix < (1LL << 31) && 3 * lab < (1LL << 31) && tiles < (1LL << 31), "ragged_plan: batch too large for 32-bit offsets");
    }
    if (table) {
        int32_t* rec = table + EXCEL_RAG_REC * B;
        rec[0] = rec[1] = 0; rec[2] = (int32_t)pix; rec[3] = (int32_t)tiles; rec[4] = (int32_t)lab; rec[5] = rec[6] = rec[7] = 0;
    }
    info->B = B; info->total_tiles = (int32_t)tiles; info->total_pix = pix; info->total_label_pix = lab; info->max_plane_pix = max_plane;
    info->table_ints = EXCEL_RAG_REC * (long long)(B + 1) + tiles;
    return EXCEL_OK;
}

static TileGeo ragged_geo(const int32_t* table, int B) {
    TileGeo g;
    g.tab = table; g.B = B; g.H = g.W = 0;
    return g;
}
static TileGeo uniform_geo(int B, int H, int W) {
    TileGeo g;
    g.tab = nullptr; g.B = B; g.H = H; g.W = W;
    return g;
}

extern "C" int excel_normalize_resize_u8_ragged(const uint8_t* hwc, const int32_t* table, int B, int S, const double* mean3, const double* std3,
                                                float* out, void* stream) {
    EXCEL_CHECK_ARG(hwc && table && out && mean3 && std3 && B > 0 && S > 0, "normalize_resize_u8_ragged: bad argument");
    return excel_launch_normalize_resize_u8_ragged(hwc, out, ragged_geo(table, B), S, mean3, std3, ST(stream));
}

extern "C" int excel_cam_upsample_bkg_ragged(const float* refined, const int32_t* ncls, const int32_t* table, const excel_ragged_info* info, int g,
                                             int Smax, float* cams, void* workspace, int flags, void* stream) {
    EXCEL_CHECK_ARG(refined && ncls && table && info && cams && workspace, "cam_upsample_bkg_ragged: null argument");
    EXCEL_CHECK_ARG((((uintptr_t)cams) & 15) == 0, "cam_upsample_bkg_ragged: cams must be 16-byte aligned");
    return excel_launch_cam_upsample_bkg_ragged(refined, ncls, (float*)workspace, cams, g, Smax, ragged_geo(table, info->B), info->total_tiles,
                                                (flags & EXCEL_CAMS_ZERO_UNUSED) ? 1 : 0, ST(stream));
}

// ------------------------------------------------------------------------------------ PAR / labels / metric
extern "C" size_t excel_par_workspace_bytes(int B, int Cmax, int H, int W, int ndil) {
    const size_t hw = (size_t)H * W;
    return align_up((size_t)B * 8 * ndil * hw * sizeof(float), 256) + align_up((size_t)B * Cmax * hw * sizeof(float), 256) +
           align_up((size_t)B * 3 * hw * sizeof(float), 256);
}

// ping-pong so that the LAST step writes `out`: out, pp alternate backwards from the end
template <class Step>
static int par_jacobi(const float* masks, float* out, float* pp, int n_iter, Step step) {
    const float* cur = masks;
    for (int it = 0; it < n_iter; ++it) {
        float* dst = (((n_iter - 1 - it) & 1) == 0) ? out : pp;
        TRY(step(cur, dst));
        cur = dst;
    }
    return EXCEL_OK;
}

extern "C" int excel_par_forward(const float* imgs, int h, int w, const float* masks, const int32_t* nchan, int B, int Cmax, int H,
                                 int W, const int32_t* dilations, int ndil, int n_iter, float w1, float w2, float* out,
                                 void* workspace, int flags, void* stream) {
    EXCEL_CHECK_ARG(imgs && masks && out && workspace && dilations, "par_forward: null argument");
    EXCEL_CHECK_ARG(n_iter >= 0 && ndil >= 1 && ndil <= 8, "par_forward: bad n_iter/ndil");
    EXCEL_CHECK_ARG(B > 0 && Cmax > 0 && H > 0 && W > 0 && h > 0 && w > 0, "par_forward: bad shape");
    const size_t hw = (size_t)H * W;
    char* base = (char*)workspace;
    float* aff = (float*)base;
    float* pp = (float*)(base + align_up((size_t)B * 8 * ndil * hw * sizeof(float), 256));
    float* guide = (float*)((char*)pp + align_up((size_t)B * Cmax * hw * sizeof(float), 256));
    hipStream_t st = ST(stream);
    const float* gimg = imgs;
    if (h != H || w != W) {   // F.interpolate(..., align_corners=True) (PAR.py:67)
        TRY(excel_launch_bilinear_ac(imgs, guide, B * 3, h, w, H, W, st));
        gimg = guide;
    }
    // Affinities: either streamed as 8*ndil planes per image (EXCEL_PAR_STREAM_AFFINITIES, or a shape / dilation set the tiled kernel
    // does not take), or recomputed in every step from the guide image and 5 per-pixel statistics: bit-identical weights from a
    // tenth of the bytes (par.hip).
    const TileGeo geo = uniform_geo(B, H, W);
    const bool recompute = !(flags & EXCEL_PAR_STREAM_AFFINITIES) && n_iter > 0 && (W % 4) == 0 && (((uintptr_t)pp & 15) == 0) &&
                           excel_par_guide_supported(gimg, aff, masks, out, Cmax, (long long)hw, W, dilations, ndil);
    TRY(excel_launch_par_affinity(gimg, aff, geo, 0, dilations, ndil, w1, w2, st, recompute ? 1 : 0));
    if (n_iter == 0) {
        hipMemcpyAsync(out, masks, sizeof(float) * (size_t)B * Cmax * hw, hipMemcpyDeviceToDevice, st);
        return EXCEL_OK;
    }
    if (recompute)
        return par_jacobi(masks, out, pp, n_iter, [&](const float* cur, float* dst) {
            return excel_launch_par_iterate_guide(gimg, aff, cur, dst, nchan, Cmax, geo, 0, dilations, ndil, w1, w2, st);
        });
    return par_jacobi(masks, out, pp, n_iter, [&](const float* cur, float* dst) {
        return excel_launch_par_iterate(aff, cur, dst, nchan, B, Cmax, H, W, dilations, ndil, st);
    });
}

// statistics [5 planes] + ping-pong [Cmax planes] + resized guide [3 planes], all in the pitched ragged layout
extern "C" size_t excel_par_ragged_workspace_bytes(long long total_pix, int Cmax) {
    return align_up((size_t)5 * total_pix * sizeof(float), 256) + align_up((size_t)Cmax * total_pix * sizeof(float), 256) +
           align_up((size_t)3 * total_pix * sizeof(float), 256);
}

extern "C" int excel_par_forward_ragged(const float* imgs, int h, int w, const float* masks, const int32_t* nchan, const int32_t* table,
                                        const excel_ragged_info* info, int Cmax, const int32_t* dilations, int ndil, int n_iter, float w1,
                                        float w2, float* out, void* workspace, void* stream) {
    EXCEL_CHECK_ARG(imgs && masks && out && workspace && dilations && table && info, "par_forward_ragged: null argument");
    EXCEL_CHECK_ARG(n_iter >= 1 && Cmax > 0 && h > 0 && w > 0 && info->B > 0, "par_forward_ragged: bad n_iter / shape");
    const size_t tp = (size_t)info->total_pix;
    char* base = (char*)workspace;
    float* stats = (float*)base;
    float* pp = (float*)(base + align_up(5 * tp * sizeof(float), 256));
    float* guide = (float*)((char*)pp + align_up((size_t)Cmax * tp * sizeof(float), 256));
    hipStream_t st = ST(stream);
    const TileGeo geo = ragged_geo(table, info->B);
    // (every pitch of the ragged layout is a multiple of 4 floats by construction - excel_ragged_plan - so 4 stands for all of them)
    EXCEL_CHECK_ARG(excel_par_guide_supported(guide, stats, masks, out, Cmax, info->max_plane_pix, 4, dilations, ndil) && (((uintptr_t)pp & 15) == 0),
                    "par_forward_ragged: needs dilations [1,2,4,8,12,24], 16-byte aligned buffers and max(Cmax,5) * H * Wp * 4 < 2^31 per image "
                    "(Wp = W rounded up to 4)");
    // the guide always goes through the align_corners=True resize (PAR.py:67): it is the identity where (H_b, W_b) == (h, w), and it
    // brings the uniform [B,3,h,w] network input into the pitched per-image layout
    TRY(excel_launch_bilinear_ac_ragged(imgs, guide, h, w, geo, info->total_tiles, st));
    TRY(excel_launch_par_affinity(guide, stats, geo, info->total_tiles, dilations, ndil, w1, w2, st, 1));
    return par_jacobi(masks, out, pp, n_iter, [&](const float* cur, float* dst) {
        return excel_launch_par_iterate_guide(guide, stats, cur, dst, nchan, Cmax, geo, info->total_tiles, dilations, ndil, w1, w2, st);
    });
}

extern "C" int excel_argmax_label_ragged(const float* cams, const int32_t* nchan, const int32_t* cls_idx, const int32_t* table,
                                         const excel_ragged_info* info, int Smax, int Cmax, uint8_t* labels_u8, void* stream) {
    EXCEL_CHECK_ARG(cams && table && info && labels_u8, "argmax_label_ragged: null argument");
    return excel_launch_argmax_label_ragged(cams, nchan, cls_idx, Smax, Cmax, ragged_geo(table, info->B), info->total_tiles, labels_u8, ST(stream));
}

extern "C" int excel_argmax_label(const float* cams, const int32_t* nchan, const int32_t* cls_idx, int B, int Smax, int Cmax,
                                  long long HW, uint8_t* labels_u8, int64_t* labels_i64, void* stream) {
    EXCEL_CHECK_ARG(cams && (labels_u8 || labels_i64), "argmax_label: null argument");
    return excel_launch_argmax_label(cams, nchan, cls_idx, B, Smax, Cmax, HW, labels_u8, (long long*)labels_i64, ST(stream));
}

extern "C" int excel_confusion_accumulate(const uint8_t* gt, const uint8_t* pred, long long n, int num_classes, int64_t* hist,
                                          void* stream) {
    EXCEL_CHECK_ARG(gt && pred && hist && n >= 0, "confusion_accumulate: null argument");
    if (n == 0) return EXCEL_OK;
    return excel_launch_confusion(gt, pred, n, num_classes, (unsigned long long*)hist, ST(stream));
}

// ------------------------------------------------------------------------------------ one-time / auxiliary
extern "C" int excel_attr_aggregate(const float* text, const float* bank, int F, int T, int C, int K, double topK, float* out,
                                    void* stream) {
    EXCEL_CHECK_ARG(text && bank && out, "attr_aggregate: null argument");
    const int drop = (int)((1.0 - topK) * (double)K);       // topk = int((1-topK) * K)  (load_attr.py:100)
    return excel_launch_attr_aggregate(text, bank, F, T, C, K, drop, out, ST(stream));
}

extern "C" int excel_bilinear_resize(const float* in, float* out, long long planes, int h, int w, int H, int W, int align_corners,
                                     void* stream) {
    EXCEL_CHECK_ARG(in && out && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bilinear_resize: bad argument");
    return excel_launch_bilinear_resize(in, out, planes, h, w, H, W, align_corners, ST(stream));
}

extern "C" int excel_pos_embed_resize(const float* pos, int side, int g, int D, float* out, void* stream) {
    EXCEL_CHECK_ARG(pos && out && side > 0 && g > 0, "pos_embed_resize: bad argument");
    hipLaunchKernelGGL(pos_resize_kernel, dim3((unsigned)cdivl((long long)(g * g + 1) * D, 256)), dim3(256), 0, ST(stream), pos, out, side, g, D);
    EXCEL_CHECK_LAUNCH("pos_resize");
    return EXCEL_OK;
}

extern "C" int excel_flip_max_normalize(const float* attr, float* out, int B, int g, int F, void* stream) {
    EXCEL_CHECK_ARG(attr && out, "flip_max_normalize: null argument");
    return excel_launch_flip_max_normalize(attr, out, B, g, F, ST(stream));
}

extern "C" int excel_lam_scale_accumulate(const float* maps, float* acc, int B, int g, int F, int H, int W, int init, void* stream) {
    EXCEL_CHECK_ARG(maps && acc && B > 0 && g > 0, "lam_scale_accumulate: bad argument");
    return excel_launch_lam_scale_accumulate(maps, acc, B, g, F, H, W, init, ST(stream));
}

extern "C" int excel_plane_minmax_normalize(float* lam, long long planes, long long HW, void* stream) {
    EXCEL_CHECK_ARG(lam && planes > 0 && HW > 0, "plane_minmax_normalize: bad argument");
    return excel_launch_plane_minmax_normalize(lam, planes, HW, ST(stream));
}
