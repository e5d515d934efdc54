// Shared device/host helpers for libexcel_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define EXCEL_OK 0
#define EXCEL_ERR_ARG (-1)
#define EXCEL_ERR_LAUNCH (-2)
#define EXCEL_ERR_ALLOC (-3)

void excel_set_error(const char* fmt, ...);

#define EXCEL_CHECK_ARG(cond, ...)                         \
    do {                                                   \
        if (!(cond)) {                                     \
            excel_set_error(__VA_ARGS__);                  \
            return EXCEL_ERR_ARG;                          \
        }                                                  \
    } while (0)

#define EXCEL_CHECK_LAUNCH(name)                                                   \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            excel_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return EXCEL_ERR_LAUNCH;                                               \
        }                                                                          \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivl(long long a, long long b) { return (a + b - 1) / b; }

#define WAVE 64

// Ablation switches exist only in development builds (EXCEL_DEV=1 python -m excel_amd.build): the shipped library reads no
// environment variable on a launch path and its kernels carry no debug branches.
#ifdef EXCEL_DEV
#define EXCEL_DBG(x) (x)
#else
#define EXCEL_DBG(x) 0
#endif

// The two halves of a wave exchange a register without an LDS round trip (v_permlane32_swap_b32, gfx950): lo = x of lane & 31, hi = x
// of 32 + (lane & 31), in every lane - max(lo, hi) / lo + hi are the cross-half reductions (__shfl_xor(x, 32) is a ds_bpermute_b32).
__device__ __forceinline__ void wave_halves(float x, float& lo, float& hi) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
// butterfly reductions; the first step (partner lane ^ 32) through the permlane swap: same pairing, same bits
__device__ __forceinline__ float wave_sum(float v) {
    float lo, hi;
    wave_halves(v, lo, hi);
    v = lo + hi;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    float lo, hi;
    wave_halves(v, lo, hi);
    v = fmaxf(lo, hi);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
    float lo, hi;
    wave_halves(v, lo, hi);
    v = fminf(lo, hi);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// C/D fragment row of a 32x32 MFMA accumulator register (guide: cdna_hip_programming.md §3)
__device__ __forceinline__ int c32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Bijective XCD-aware remap of a linear workgroup id: the hardware places id b on XCD b % 8;
// give every XCD a contiguous chunk of the logical tile order so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, loc = id >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

// ---------------------------------------------------------------- launch geometry: uniform batch or ragged batch (tile map)
// Ragged batches (include/excel_hip.h, "ragged batches"): B images of different sizes in one launch.  Planes are PITCHED (row pitch
// Wp = W rounded up to 4 floats, so every row and plane starts 16-byte aligned), image b of a K-plane tensor starts at element
// K * poff_b; u8 label maps are tight, image b at loff_b.  Work is cut into 64 x 16 pixel tiles numbered image-major, row-major;
// `tab` (device, int32) = (B + 1) records of 8 ints {H, W, poff, tile_off, loff, 0, 0, 0} (record B holds the totals) followed by the
// image index of every tile.  tab == nullptr: a uniform batch of [H, W] planes without padding (Wp = W).
#define EXCEL_RAG_REC 8
struct TileGeo {
    const int* tab;      // ragged table, or nullptr
    int B;
    int H, W;            // uniform geometry (tab == nullptr)
};
struct Tile {
    int b, x0, y0, H, W, Wp;
    long long base;      // element offset of image b in a ONE-plane tensor (multiply by the plane count of the tensor)
    long long HW;        // plane stride of image b
    long long lab;       // offset of image b in a tight u8 label tensor
};
// blockIdx -> tile.  Uniform launches use grid (cdiv(W,64), cdiv(H,16), B); ragged launches grid.x = total tiles.  All values are
// wave-uniform (scalar loads / SALU only).
__device__ __forceinline__ Tile tile_of_ragged(const TileGeo& g, int tile) {     // tile = index into the ragged tile list
    Tile t;
    const int* __restrict__ tab = g.tab;
    t.b = tab[EXCEL_RAG_REC * (g.B + 1) + tile];
    const int* rec = tab + EXCEL_RAG_REC * t.b;
    t.H = rec[0]; t.W = rec[1];
    t.Wp = (t.W + 3) & ~3;
    t.base = rec[2];
    t.lab = rec[4];
    const int loc = tile - rec[3];
    const int ntx = (t.W + 63) >> 6;
    const int ty = loc / ntx;
    t.x0 = (loc - ty * ntx) * 64;
    t.y0 = ty * 16;
    t.HW = (long long)t.H * t.Wp;
    return t;
}
template <bool RAGGED>
__device__ __forceinline__ Tile tile_of(const TileGeo& g) {
    Tile t;
    if (RAGGED) {
        t = tile_of_ragged(g, blockIdx.x);
    } else {
        t.b = blockIdx.z; t.x0 = blockIdx.x * 64; t.y0 = blockIdx.y * 16;
        t.H = g.H; t.W = g.W; t.Wp = g.W;
        t.HW = (long long)g.H * g.W;
        t.base = (long long)t.b * t.HW;
        t.lab = t.base;
    }
    return t;
}

// ---------------------------------------------------------------- optional per-category HIP-event profiling
// (bench.py enables it to get live per-kernel durations on the launch stream; off by default: zero overhead)
enum ExcelProfCat {
    PROF_GEMM_NT = 0, PROF_GEMM_NN, PROF_GEMM_BF16X3, PROF_ATTN_ROWPASS, PROF_ATTN_ACCUM, PROF_LAYERNORM, PROF_EMBED, PROF_TOKEN_NORM,
    PROF_CAM_EPILOGUE, PROF_SINKHORN, PROF_BBOX, PROF_MATVEC, PROF_UPSAMPLE, PROF_PAR_AFFINITY, PROF_PAR_ITERATE,
    PROF_ARGMAX, PROF_CONFUSION, PROF_OTHER, PROF_CAM_PROJ, PROF_CAM_FUSED, PROF_NCAT
};
extern thread_local int g_excel_prof_gemm_cat;               // >= 0: the next GEMM launches are booked under this category (the CAM's projection GEMM)
extern bool g_excel_prof_on;
extern unsigned long long g_excel_prof_mask;   // bit c set: category c is bracketed with events
extern int g_excel_prof_every;                 // bracket every n-th launch of a category (an event pair costs ~10 us of GPU idle)
extern unsigned g_excel_prof_seen[];
void excel_prof_begin(int cat, hipStream_t st, double work);
void excel_prof_end(int cat, hipStream_t st);
struct ProfScope {
    int cat; hipStream_t st; bool on;
    ProfScope(int c, hipStream_t s, double work = 0.0)
        : cat(c), st(s), on(g_excel_prof_on && ((g_excel_prof_mask >> c) & 1) && (g_excel_prof_seen[c]++ % g_excel_prof_every) == 0) { if (on) excel_prof_begin(cat, st, work); }
    ~ProfScope() { if (on) excel_prof_end(cat, st); }
};


// ---------------------------------------------------------------- LDS reads that the compiler's wait-count pass does not see
// The 16-bit type of the "split" operand planes (x = hi + lo, three MFMAs per product).  bf16 (default): 8 + 8 mantissa bits at the fp32
// exponent range - no overflow or underflow concerns.  EXCEL_SPLIT_F16 (build-time experiment, `EXCEL_SPLIT_F16=1 python -m
// excel_amd.build`): IEEE half, 11 + 11 bits where lo stays a normal half (|x| >= 2^-3; fewer below: lo goes denormal under 6e-5), range
// 65 504 - same MFMA rate (v_mfma_f32_32x32x16_f16), same layouts.
#ifdef EXCEL_SPLIT_F16
typedef _Float16 split_t;
#define EXCEL_MFMA16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
#define EXCEL_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define EXCEL_SPLIT_NAME "f16"
#else
typedef __bf16 split_t;
#define EXCEL_MFMA16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)
// 16x16x32: the same flops per cycle, K = 32 inside one instruction - half the accumulator read-modify-writes per flop.  Under the chip's
// power cap that is the cheaper form: the GEMM's MFMA stream alone sustains 2 127 TFLOP/s with it against 1 800 with 32x32x16 on random
// operand bits (tools_dev/micro/mfma_power.hip, profiles/r04_micro_mfma_power.txt).  A: row = lane % 16, k = 8 (lane / 16) + 0..7;
// B: column = lane % 16, same k; D: row = 4 (lane / 16) + reg, column = lane % 16.
#define EXCEL_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define EXCEL_SPLIT_NAME "bf16"
#endif
typedef split_t splitx8 __attribute__((ext_vector_type(8)));
typedef split_t splitx4 __attribute__((ext_vector_type(4)));
// fp32 -> one 16-bit plane (hi = split_hi(x), lo = split_hi(x - hi)): a plain round-to-nearest conversion for both types.  IEEE half has a
// range of 65 504: beyond it hi becomes +-inf, lo = x - hi the opposite infinity, and every product they enter is a NaN - an overflow of
// the f16 modes is LOUD (check_numerics compares NaN as a failure and moves to exact fp32), never a finite wrong number; a NaN input
// stays NaN in both planes; bf16 has the fp32 range.  (Rounds 4-5 saturated instead - v_med3 + a NaN test per value, 18 VALU
// operations per pair of values against 6 - which made every f16 epilogue and the f16 row pass measurably slower than their bf16 twins;
// none of the split tensors of the path - LayerNorm / GELU outputs, q|k|v, probabilities, A_sum, weights - comes near the range.)
__device__ __forceinline__ split_t split_hi(float x) { return (split_t)x; }
// Two values at once -> packed planes (low 16 bits = a's plane value, high 16 bits = b's): hi2 = {hi(a), hi(b)}, lo2 = {lo(a), lo(b)}.  bf16:
// five instructions for the pair (v_cvt_pk_bf16_f32, shift, mask, v_pk_add_f32, v_cvt_pk_bf16_f32) and the result is already in
// store order - the element-at-a-time form costs ~5.5 instructions per element plus a v_perm to pack.  Same bits as split_hi.
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi2, unsigned& lo2) {
#ifdef EXCEL_SPLIT_F16
    typedef float f2_ __attribute__((ext_vector_type(2)));
    typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
    const f2_ v = {a, b};
    const h2_ h = __builtin_convertvector(v, h2_);
    hi2 = __builtin_bit_cast(unsigned, h);
    lo2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v - __builtin_convertvector(h, f2_), h2_));
#else
    typedef float f2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
    const f2_ v = {a, b};
    hi2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2_));
    const f2_ back = {__uint_as_float(hi2 << 16), __uint_as_float(hi2 & 0xffff0000u)};
    lo2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v - back, b2_));
#endif
}
// Split of values known to be FINITE AND INSIDE the 16-bit type's range (softmax probabilities): hi = the leading bits of x cut with a bit
// mask - exactly representable, so its conversion is exact in every rounding mode and needs neither the saturation nor the NaN test of
// split_hi - and lo = rne(x - hi) (the difference is exact in fp32).  Two values -> packed planes.  IEEE half: 11 + 11 significant bits
// (mask 0xFFFFE000, v_cvt_pkrtz_f16_f32 for the pair); bf16: 8 + 8 (mask 0xFFFF0000).  ~3.5 VALU operations per value; the checked
// split_hi form costs 10 for IEEE half (round 6: it was what made the f16 row pass 17 % slower than the bf16 one).
__device__ __forceinline__ void split_pair_bounded(float a, float b, unsigned& hi2, unsigned& lo2) {
#ifdef EXCEL_SPLIT_F16
    typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
    const float ha = __uint_as_float(__float_as_uint(a) & 0xFFFFE000u), hb = __uint_as_float(__float_as_uint(b) & 0xFFFFE000u);
    hi2 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ha, hb));
    const h2_ l = {(_Float16)(a - ha), (_Float16)(b - hb)};
    lo2 = __builtin_bit_cast(unsigned, l);
#else
    typedef float f2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    hi2 = (ua >> 16) | (ub & 0xFFFF0000u);
    const f2_ d = {a - __uint_as_float(ua & 0xFFFF0000u), b - __uint_as_float(ub & 0xFFFF0000u)};
    lo2 = __builtin_bit_cast(unsigned, __builtin_convertvector(d, b2_));
#endif
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// QuickGELU x * sigmoid(1.702 x) (clip/clip_surgery_model.py QuickGELU) over NV register tiles of four values, written stage by stage with
// packed fp32 operations: exactly the operations and roundings of  x * rcp(1 + __expf(-1.702f * x))  ( t = x * -1.702, u = t * log2(e),
// e = exp2(u), d = 1 + e, r = rcp(d), x * r ) - every GEMM instance produces the same bits - but as 2 NV independent chains side by side.
// The element-at-a-time form compiles to ONE serial chain through one temporary (mul, mul, exp, nop, add, rcp per value): on the four-wave
// GEMM, one wave per SIMD, nothing fills its latencies - it was the largest part of the fc1 epilogue (round 5, read off the disassembly).
template <int NV>
__device__ __forceinline__ void quickgelu_tiles(f32x4* v) {       // v[0 .. NV)
    f32x2 t[2 * NV];
#pragma unroll
    for (int i = 0; i < 2 * NV; ++i) t[i] = f32x2{v[i >> 1][2 * (i & 1)], v[i >> 1][2 * (i & 1) + 1]} * f32x2{-1.702f, -1.702f};
#pragma unroll
    for (int i = 0; i < 2 * NV; ++i) t[i] = t[i] * f32x2{1.4426950408889634f, 1.4426950408889634f};
#pragma unroll
    for (int i = 0; i < 2 * NV; ++i) t[i] = f32x2{__builtin_amdgcn_exp2f(t[i][0]), __builtin_amdgcn_exp2f(t[i][1])};
#pragma unroll
    for (int i = 0; i < 2 * NV; ++i) t[i] = t[i] + f32x2{1.f, 1.f};
#pragma unroll
    for (int i = 0; i < 2 * NV; ++i) t[i] = f32x2{__builtin_amdgcn_rcpf(t[i][0]), __builtin_amdgcn_rcpf(t[i][1])};
#pragma unroll
    for (int i = 0; i < 2 * NV; ++i) {
        const f32x2 o = f32x2{v[i >> 1][2 * (i & 1)], v[i >> 1][2 * (i & 1) + 1]} * t[i];
        v[i >> 1][2 * (i & 1)] = o[0];
        v[i >> 1][2 * (i & 1) + 1] = o[1];
    }
}
// LDS accesses of the streaming loop are written as inline asm: the compiler's wait-count pass treats every ds_read as a
// possible reader of a pending global_load_lds and drains the whole DMA queue (s_waitcnt vmcnt(0)) in front of it, which
// serialises the tile stream (that is what held attn_accum_bf_kernel at ~20 % matrix-core busy).  The waits here are explicit.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ splitx8 lds_read16(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_bit_cast(splitx8, v);
}
__device__ __forceinline__ float2 lds_read8(unsigned addr) {
    float2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write8(unsigned addr, float2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// s_waitcnt lgkmcnt(0) that the consumers of the eight fragments depend on (keeps the MFMAs behind the wait)
__device__ __forceinline__ splitx4 lds_read8h(unsigned addr) {
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_bit_cast(splitx4, v);
}
// hardware transpose read (gfx950): the 16 lanes of a group each address one 8-byte row segment of a [4 rows][16 columns] block of
// 16-bit elements (segment i = row i / 4, columns 4 (i % 4) ..+3, any row stride); lane c of the group receives column c, rows 0..3
template <int OFF>
__device__ __forceinline__ splitx4 lds_read8h_tr(unsigned addr) {      // address + compile-time byte offset (the instruction's immediate)
    static_assert(OFF >= 0 && OFF < 65536, "ds_read_b64_tr_b16 immediate offset");
    f32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return __builtin_bit_cast(splitx4, v);
}
__device__ __forceinline__ void lds_wait8(splitx8 (&a)[4], splitx8 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])::"memory");
}


// ---------------------------------------------------------------- split-bf16 operand layout
// A "split" row of K fp32 values is 2K bf16: for every block of 32 k the 32 hi values then the 32 lo values, so that the
// 128 bytes a bf16x3 GEMM k-step (BK = 32) needs from a row are ONE contiguous, 128-B aligned cache line.
__host__ __device__ __forceinline__ long long split_off(int k, int plane) { return (long long)(k >> 5) * 64 + plane * 32 + (k & 31); }
