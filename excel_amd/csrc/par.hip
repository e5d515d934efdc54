// Pixel-adaptive refinement (utils/PAR.py:26-92) + arg-max labelling (utils/affutils.py:80-89) +
// confusion-matrix accumulation (utils/evaluate.py:9-20).
//
//   par_affinity       : guide image -> per pixel: 48 edge-clamped dilated taps per channel, unbiased std over the taps,
//                        -(|I_nb - I|/(std+1e-8)/w1)^2, mean over RGB, softmax over taps + w2 * softmax(position term)   (PAR.py:70-86)
//                        written either as the 5 statistics the 48 weights are a function of (COMPACT) or as 48 planes
//   par_iterate_guide  : masks'[c] = sum_t masks[c][nb_t] * aff[t]  (Jacobi step, ping-pong buffers; :88-90) with the 48 weights
//                        RECOMPUTED from the guide image + the statistics: the production kernel, uniform and ragged batches
//   par_iterate_stream : the same step from 48 streamed planes, any shape / dilation set: the fall-back for shapes the tiled kernel
//                        does not take (W % 4 != 0 in the un-pitched uniform layout, other dilation sets) and the bit-exactness
//                        reference of the recomputing kernel (same operations in the same order -> same bits)
//   argmax_label       : label = valid_key[argmax_c]                                      (affutils.py:86-87)
//   confusion          : hist[nc*gt + pred] += 1 over gt < nc                             (evaluate.py:10-14)
//
// Algorithmic traffic of one step = (48 + 2C) * H * W * 4 bytes; the recomputing kernel moves (5 + 3 + 2C) * 4 B/pixel.
// The exponent is evaluated in base 2: log2(e)/3 (the RGB mean) is folded into the per-channel scale k2_c, so one tap costs
// sub, mul, fma per guide channel + sub, v_exp_f32, fma - no separate multiplications by 1/3 and log2(e).
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "common.h"
#include "excel_internal.h"

struct ParDil {
    int d[8];
    float pos_sm[64];   // w2 * softmax over taps of the position term (constant vector, PAR.py:83,86)
};

#define PAR_LOG2E_3 0.48089834696298783f      // log2(e) / 3
#ifndef PAR_MASK_DEPTH
#define PAR_MASK_DEPTH 1                       // tap rows in flight in the mask phase of par_iterate_guide_kernel
#endif

// COMPACT: instead of the 48 affinity planes write the 5 per-pixel statistics they are a function of,
//   stats[0..2] = k2_c = log2(e)/3 / ((std_c + 1e-8) w1)^2,   stats[3] = m (max of the 48 base-2 exponents),   stats[4] = 1 / sum 2^(z - m)
// par_iterate_guide_kernel recomputes  aff_t = 2^(sum_c -(I_nb - I)^2 k2_c - m) / sum + pos_t  from them with the SAME operations
// in the same order, i.e. bit-identical weights, from 20 B/pixel instead of 192 B/pixel.
// (a RUN-TIME flag on one instantiation: both output forms come from the same compiled arithmetic.)
// Uniform launch: grid (cdiv(W,64), cdiv(H,4), B); ragged launch: grid (total 64x16 tiles, 4 row groups).
template <int ND, bool RAGGED>
__global__ __launch_bounds__(256) void par_affinity_kernel(const float* __restrict__ img, float* __restrict__ aff, ParDil dl, TileGeo geo,
                                                           float w1, int COMPACT) {
    constexpr int NT = 8 * ND;
    int b, H, W, Wp, x, y;
    long long base, HW;
    // one wave = one image row: y and every tap row are wave-uniform -> scalar row bases, per-lane work is the x offset only
    // (per-lane 64-bit address arithmetic for the 144 tap loads was most of this kernel's VALU work)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (RAGGED) {
        const Tile t = tile_of<true>(geo);
        b = t.b; H = t.H; W = t.W; Wp = t.Wp; base = t.base; HW = t.HW;
        x = t.x0 + (threadIdx.x & 63);
        y = t.y0 + blockIdx.y * 4 + wave;
    } else {
        b = blockIdx.z; H = geo.H; W = geo.W; Wp = geo.W; HW = (long long)H * W; base = (long long)b * HW;
        x = blockIdx.x * 64 + (threadIdx.x & 63);
        y = blockIdx.y * 4 + wave;
    }
    if (y >= H) return;
    const bool in_x = x < W;
    const int xc = min(x, W - 1);
    int xo[ND][2];
#pragma unroll
    for (int di = 0; di < ND; ++di) { xo[di][0] = max(xc - dl.d[di], 0); xo[di][1] = min(xc + dl.d[di], W - 1); }
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
    const long long pix = (long long)y * Wp + x;
    float* st = aff + 5 * base + pix;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const float* ch = img + 3 * base + c * HW;
        const float* row0 = ch + (long long)y * Wp;
        const float ctr = row0[xc];
        float nb[NT];
        float sum = 0.f;
#pragma unroll
        for (int di = 0; di < ND; ++di) {
            const int d = dl.d[di];
            const float* rup = ch + (long long)max(y - d, 0) * Wp;          // scalar
            const float* rdn = ch + (long long)min(y + d, H - 1) * Wp;
            // tap order of get_dilated_neighbors (PAR.py:39-52): (-1,-1) (-1,0) (-1,1) (0,-1) (0,1) (1,-1) (1,0) (1,1)
            nb[di * 8 + 0] = rup[xo[di][0]]; nb[di * 8 + 1] = rup[xc]; nb[di * 8 + 2] = rup[xo[di][1]];
            nb[di * 8 + 3] = row0[xo[di][0]]; nb[di * 8 + 4] = row0[xo[di][1]];
            nb[di * 8 + 5] = rdn[xo[di][0]]; nb[di * 8 + 6] = rdn[xc]; nb[di * 8 + 7] = rdn[xo[di][1]];
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += nb[di * 8 + k];
        }
        const float mean = sum / (float)NT;
        float var = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) { const float dv = nb[t] - mean; var = fmaf(dv, dv, var); }      // (explicit: par_stats_tile_kernel must round alike)
        const float den = sqrtf(var / (float)(NT - 1)) + 1e-8f;     // unbiased std (torch.std default)
        // (|nb - ctr| / den / w1)^2 / 3 * log2(e) as one fma per tap: 3 true divisions per pixel instead of 2 x 144 (the divisions
        // were ~80 % of this kernel's instructions); differs from the literal form by two roundings (~2e-7 relative)
        const float k = 1.f / (den * w1);
        const float k2 = __fmul_rn(k * k, PAR_LOG2E_3);
        if (COMPACT && in_x) st[c * HW] = k2;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float dv = nb[t] - ctr;
            acc[t] = fmaf(-(dv * dv), k2, acc[t]);
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) m = fmaxf(m, acc[t]);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t] = __builtin_amdgcn_exp2f(__fsub_rn(acc[t], m)); s += acc[t]; }
    const float inv_s = 1.f / s;
    if (!in_x) return;
    if (COMPACT) {
        st[3 * HW] = m;
        st[4 * HW] = inv_s;
        return;
    }
    float* out = aff + (long long)NT * base + pix;
#pragma unroll
    for (int t = 0; t < NT; ++t) out[(long long)t * HW] = fmaf(acc[t], inv_s, dl.pos_sm[t]);
}

// Jacobi step from streamed affinity planes, any shape, any 1..8 dilations (one instantiation, run-time loops): not a production
// kernel.  Fused multiply-adds in tap order, like the recomputing kernel.
#define PAR_CCH 8
__global__ __launch_bounds__(256) void par_iterate_stream_kernel(const float* __restrict__ aff, const float* __restrict__ in,
                                                                 float* __restrict__ out, const int* __restrict__ nchan, ParDil dl,
                                                                 int ndil, int Cmax, int H, int W) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int nch = nchan ? min(nchan[b], Cmax) : Cmax;
    const long long HW = (long long)H * W;
    const float* a = aff + (long long)b * 8 * ndil * HW + (long long)y * W + x;
    const float* ib = in + (long long)b * Cmax * HW;
    float* ob = out + (long long)b * Cmax * HW + (long long)y * W + x;
    const int dils[8] = {dl.d[0], dl.d[1], dl.d[2], dl.d[3], dl.d[4], dl.d[5], dl.d[6], dl.d[7]};
    for (int c0 = 0; c0 < nch; c0 += PAR_CCH) {
        float acc[PAR_CCH];
#pragma unroll
        for (int c = 0; c < PAR_CCH; ++c) acc[c] = 0.f;
#pragma unroll 1
        for (int di = 0; di < ndil; ++di) {
            int d = dils[0];                // select chain with static indices: a dynamically indexed kernarg struct goes to scratch
#pragma unroll
            for (int q = 1; q < 8; ++q) d = (di == q) ? dils[q] : d;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int dy = (k < 3) ? -1 : (k < 5 ? 0 : 1);
                const int dx = (k == 0 || k == 3 || k == 5) ? -1 : ((k == 1 || k == 6) ? 0 : 1);
                const int yy = min(max(y + dy * d, 0), H - 1);
                const int xx = min(max(x + dx * d, 0), W - 1);
                const float wgt = a[(long long)(di * 8 + k) * HW];
                const float* src = ib + (long long)yy * W + xx;
#pragma unroll
                for (int c = 0; c < PAR_CCH; ++c)
                    if (c0 + c < nch) acc[c] = fmaf(src[(long long)(c0 + c) * HW], wgt, acc[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < PAR_CCH; ++c)
            if (c0 + c < nch) ob[(long long)(c0 + c) * HW] = acc[c];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Jacobi step that RECOMPUTES the affinities from the guide image (par_affinity_kernel COMPACT statistics).
// The 48 weights of a pixel are a function of 3 image values per tap and 5 per-pixel statistics; streaming them as 48 fp32
// planes made the step HBM-bound at (48 + 2C) x 4 B/pixel - here a step reads (5 + 3 + 2C) x 4 B/pixel and pays
// ~8 VALU operations per tap.  One workgroup = a 64 x 16 pixel tile, 512 threads, thread = 2 pixels; the planes it needs - the
// three guide channels, then the image's mask channels - go through a DOUBLE-BUFFERED LDS tile (64 rows x 128 floats incl. the
// 24-pixel halo), filled by LDS-DMA (16-byte pieces for interior tiles, 4-byte pieces with edge-clamped per-lane source addresses =
// replicate padding at the left / right border): plane p+1 streams in while the taps of plane p are evaluated, one barrier per plane,
// no staging registers.  Phase 1 (guide planes) accumulates the tap exponents in wall[6][8] (float2 = this thread's 2 pixels), a
// finalisation turns them into the weights, phase 2 (mask planes) applies them.
// Same operations in the same order as par_affinity_kernel + the streamed-plane kernel -> bit-identical results.
// RAGGED: the tile comes from the tile map, rows are pitched (Wp = W rounded up to 4): any width works, the (up to 3) padding columns
// of a row are computed and stored like pixels and never read as neighbours (border tiles clamp their source columns to W - 1).
//
// Thread = 2 pixels keeps the 96 weight registers + everything else under 128 VGPRs -> 4 waves per SIMD, two workgroups per CU (round
// 2's float4 form held 192 weights in 256 VGPRs: 2 waves per SIMD, and every tap row waited out its own LDS round trip - the mask
// phase issued 6 packed fma per ~130 cycles; measured 4.47 -> 3.39 ms per 32-image step, same bits).  The dilation set is a compile-time
// constant (par_dil_ok admits nothing else), so every tap address is ONE per-thread base register + an immediate offset: no per-tap
// address arithmetic, and tap row g+1 is in flight while row g is consumed (counted lgkmcnt).
#define PG_TP 128            // LDS row pitch (floats): 512 B keeps every ds_read group on distinct banks
template <int DI> struct ParD { static constexpr int v = DI == 0 ? 1 : DI == 1 ? 2 : DI == 2 ? 4 : DI == 3 ? 8 : DI == 4 ? 12 : 24; };
template <class F, int... I> __device__ __forceinline__ void static_for_seq(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_seq(f, std::make_integer_sequence<int, N>{}); }
template <int OFF> __device__ __forceinline__ f32x2 lds_read8_imm(unsigned base) {
    static_assert(OFF >= 0 && OFF < 65536 && (OFF & 7) == 0, "ds_read_b64 immediate offset");
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF) : "memory");
    return v;
}

template <bool RAGGED>
__global__ __launch_bounds__(512, 4) void par_iterate_guide_kernel(const float* __restrict__ guide, const float* __restrict__ stats,
                                                                    const float* __restrict__ in, float* __restrict__ out,
                                                                    const int* __restrict__ nchan, ParDil dl, int Cmax, TileGeo geo, int dbg) {
    constexpr int ND = 6, HALO = 24, TR = 16 + 2 * HALO, TP = PG_TP, NG = 3 * ND;
    __shared__ __attribute__((aligned(1024))) float tile[2 * TR * TP];   // [2][TR][TP]
    // XCD-aware tile order (round 4): the dispatcher deals workgroups round-robin over the 8 XCDs in linear grid order, so neighbouring
    // tiles - which share 3/4 of their staged halo - never met in one L2.  XCD x now owns a contiguous eighth of the tile list (whole
    // images of a uniform batch): fabric fetch per launch 880 -> 274 MB (= the planes once; rocprofv3 FETCH_SIZE), time unchanged
    // (3.87 vs 3.87 ms same-box: this kernel is not bound by where its bytes come from).  A pure permutation of the tiles.
    Tile tg;
    if (RAGGED) {
        tg = tile_of_ragged(geo, (EXCEL_DBG(dbg) & 32) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x));
    } else {
        tg = tile_of<false>(geo);
        if (!(EXCEL_DBG(dbg) & 32)) {                            // (dev arm bit 5: the plain grid order)
            const int nx = gridDim.x, ny = gridDim.y, per = nx * ny;
            const int id = xcd_remap(blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z), per * (int)gridDim.z);
            const int bb = id / per, rem = id - bb * per, tyy = rem / nx;
            tg.b = bb; tg.x0 = (rem - tyy * nx) * 64; tg.y0 = tyy * 16; tg.base = (long long)bb * tg.HW; tg.lab = tg.base;
        }
    }
    const int b = tg.b, x0 = tg.x0, y0 = tg.y0, H = tg.H, W = tg.W, Wp = tg.Wp;
    const long long HW = tg.HW;
    const int tid = threadIdx.x, lane = tid & 63, tx = tid & 31, ty = tid >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = x0 + 2 * tx, py = y0 + ty;
    const int nch = nchan ? min(nchan[b], Cmax) : Cmax;
    const long long pix = (long long)min(py, H - 1) * Wp + min(px, Wp - 2);
    const float* st_b = stats + 5 * tg.base + pix;

    // ---- plane staging by LDS-DMA: wave w fills rows [w*TR/8, (w+1)*TR/8) of the tile
    typedef __attribute__((address_space(3))) unsigned char* lds_bptr;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(guide + 3 * tg.base), 0, (int)(3 * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (long long)Cmax * tg.base), 0, (int)((long long)Cmax * HW * 4), 0x00020000);
    const int xoffA = min(max(x0 - HALO + lane, 0), W - 1) * 4, xoffB = min(max(x0 - HALO + 64 + lane, 0), W - 1) * 4;   // replicate padding in x
    const unsigned tile_b = lds_addr(tile);
    constexpr int RPW = TR / 8;                                  // rows per wave
#ifdef PAR_ALL_INTERIOR      // timing-only build (tools_dev): every tile takes the 16-byte path - what padded planes would buy the 2 of 7 border tiles of a 448-wide row
    const bool interior = true;
#else
    const bool interior = x0 - HALO >= 0 && x0 + 64 + HALO <= W;
#endif
    const int x4off = min(x0 - HALO + 4 * (lane & 31), Wp - 4) * 4;         // (columns >= 64 + 2 HALO of the 128-float row are never read)
    auto stage = [&](int p, int buf) {                           // plane p of the sequence [guide 0..2, mask 0..nch-1] -> buffer buf
        const bool is_g = p < 3;
        const int plane_off = (is_g ? p : p - 3) * (int)(HW * 4);
        unsigned dst = tile_b + (buf * TR + wave * RPW) * (TP * 4);
        asm volatile("" : "+s"(dst));
        // dev arm (timing only, wrong results): stage 48 of the 64 rows = roughly the +-12-pixel apron of the round-3 review's proposal
        if ((EXCEL_DBG(dbg) & 8) && (wave == 0 || wave == 7)) return;
        if (interior) {
            int half = lane >> 5;                                // opaque: the per-lane source offsets are recomputed per plane (hoisted out of
            asm volatile("" : "+v"(half));                       // the plane loops they were live across the weight registers and spilled)
#pragma unroll
            for (int j = 0; j < RPW / 2; ++j) {
                const int gy = min(max(y0 - HALO + wave * RPW + 2 * j + half, 0), H - 1);   // replicate padding in y (per half-wave)
                const int voff = gy * Wp * 4 + x4off;
                if (is_g) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_bptr)(unsigned long long)(dst + j * (2 * TP * 4)), 16, voff, plane_off, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m, (lds_bptr)(unsigned long long)(dst + j * (2 * TP * 4)), 16, voff, plane_off, 0, 0);
            }
            return;
        }
#pragma unroll 4
        for (int rr = 0; rr < RPW; ++rr) {
            const int gy = min(max(y0 - HALO + wave * RPW + rr, 0), H - 1);                    // replicate padding in y (scalar)
            const int soff = plane_off + gy * Wp * 4;
            if (is_g) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_bptr)(unsigned long long)(dst + rr * (TP * 4)), 4, xoffA, soff, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_bptr)(unsigned long long)(dst + rr * (TP * 4) + 256), 4, xoffB, soff, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m, (lds_bptr)(unsigned long long)(dst + rr * (TP * 4)), 4, xoffA, soff, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m, (lds_bptr)(unsigned long long)(dst + rr * (TP * 4) + 256), 4, xoffB, soff, 0, 0);
            }
        }
    };

    f32x2 wall[ND][8];
#pragma unroll
    for (int di = 0; di < ND; ++di)
#pragma unroll
        for (int k = 0; k < 8; ++k) wall[di][k] = f32x2{0.f, 0.f};

    // taps of one staged plane: FN(di, k, float2 of the 2 neighbours) in tap order.  Tap row g = (dilation g/3, row -d / 0 / +d) is
    // 3 (2 for the centre row) ds_read_b64 from `base` = the thread's tile origin in the plane's buffer; row g+1 is issued before
    // row g is consumed.  d = 1 reads the aligned pairs around the pixel pair and shifts.
    const unsigned org = tile_b + (ty * TP + 2 * tx) * 4;
    // DEPTH = tap rows in flight behind the one being consumed.  The guide phase (24 VALU per row) runs at 1; the mask phase has 3 fused
    // multiply-adds per row against an LDS round trip per row, so it keeps DEPTH rows ahead (a ring of DEPTH + 1 register rows)
    auto taps = [&](auto depth_c, int buf, auto&& fn, auto&& pin) {
        constexpr int DEPTH = decltype(depth_c)::value, NS = DEPTH + 1;
        unsigned base = org + buf * (TR * TP * 4);
        asm volatile("" : "+v"(base));
        f32x2 sl[NS][3];
        auto nreads = [](int G) constexpr { return (G % 3 != 1 || G / 3 == 0) ? 3 : 2; };          // the centre row of a dilation > 1 has no middle read
        auto issue = [&](auto g) {
            constexpr int G = decltype(g)::value, r = G % 3, d = ParD<G / 3>::v, s = G % NS;
            constexpr int row = (HALO + (r - 1) * d) * TP * 4, cl = (d == 1) ? HALO - 2 : HALO - d, cr = (d == 1) ? HALO + 2 : HALO + d;
            sl[s][0] = lds_read8_imm<row + cl * 4>(base);
            if constexpr (r != 1 || d == 1) sl[s][1] = lds_read8_imm<row + HALO * 4>(base);
            sl[s][2] = lds_read8_imm<row + cr * 4>(base);
        };
        static_for<DEPTH>([&](auto g) { issue(g); });
        static_for<NG>([&](auto g) {
            constexpr int G = decltype(g)::value, di = G / 3, r = G % 3, d = ParD<di>::v, s = G % NS;
            constexpr bool mid = (r != 1 || d == 1);
            if constexpr (G + DEPTH < NG) issue(std::integral_constant<int, G + DEPTH>{});
            // reads issued behind row G's: rows G+1 .. min(G + DEPTH, NG - 1) may stay in flight
            constexpr int last = (G + DEPTH < NG) ? G + DEPTH : NG - 1;
            constexpr int nxt = [&]() constexpr { int n = 0; for (int q = G + 1; q <= last; ++q) n += nreads(q); return n; }();
            static_assert(nxt <= 15, "lgkmcnt is a 4-bit counter");
            if constexpr (mid) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(sl[s][0]), "+v"(sl[s][1]), "+v"(sl[s][2]) : "n"(nxt) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(sl[s][0]), "+v"(sl[s][2]) : "n"(nxt) : "memory");
            constexpr int k0 = (r == 0) ? 0 : (r == 1 ? 3 : 5);
            const f32x2 L = sl[s][0], R = sl[s][2];
            if constexpr (d == 1) {
                const f32x2 M = sl[s][1];
                fn(di, k0, f32x2{L[1], M[0]});
                if constexpr (r != 1) fn(di, k0 + 1, M);
                fn(di, k0 + (r != 1 ? 2 : 1), f32x2{M[1], R[0]});
            } else {
                fn(di, k0, L);
                if constexpr (r != 1) fn(di, k0 + 1, sl[s][1]);
                fn(di, k0 + (r != 1 ? 2 : 1), R);
            }
            pin();                                               // row g is consumed HERE, between the reads of the rows behind it
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // dev arm (timing only): what reading the eight d = 24 taps of a plane from global / L2 instead of LDS would ADD - 8 edge-clamped
    // float2 loads per thread per plane, summed into the result
    auto far_taps = [&](const float* plane) -> f32x2 {
        f32x2 sum = {0.f, 0.f};
        const int yu = max(py - 24, 0), yd = min(py + 24, H - 1), ym = min(py, H - 1);
        const int xl = max(min(px, Wp - 2) - 24, 0), xr = min(min(px, Wp - 2) + 24, Wp - 2), xm = min(px, Wp - 2);
        const int ys[8] = {yu, yu, yu, ym, ym, yd, yd, yd}, xs[8] = {xl, xm, xr, xl, xr, xl, xm, xr};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 v = *reinterpret_cast<const float2*>(plane + (long long)ys[k] * Wp + (xs[k] & ~1));
            sum += f32x2{v.x, v.y};
        }
        return sum;
    };
    const int np = 3 + nch;
    // The per-pixel statistics (k2_r, k2_g, k2_b, m, 1/sum) come in ONE PLANE AHEAD, issued in front of the LDS-DMA pieces of the next plane:
    // vmcnt retires in order, so a load issued behind those pieces - where the value is needed - made its own wait a wait for the whole
    // next plane, and the plane pipeline did not overlap with the taps at all (round 5, read off the disassembly: `global_load` of k2,
    // then `s_waitcnt vmcnt(0)` in the first tap row).  Issued here, it is covered by the wait the plane barrier needs anyway.  The
    // loads stay compiler-visible (an asm load whose result is still in flight gets copied / its register reused above the wait: tried,
    // memory fault); a schedule barrier keeps them in front of the DMA builtins.
    auto stat_load = [&](int plane) -> f32x2 { return *reinterpret_cast<const f32x2*>(st_b + (long long)plane * HW); };
    f32x2 st_nxt = stat_load(0), st_nx2 = f32x2{0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    stage(0, 0);
#pragma unroll 1
    for (int p = 0; p < 3; ++p) {
        // this wave's pieces of plane p and its statistics have landed.  (The statistics pass through the wait as an operand: the compiler
        // then waits for them HERE - its own count at the first use knows nothing of the DMA pieces issued in between: vmcnt(1), measured)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(st_nxt) :: "memory");
        __builtin_amdgcn_s_barrier();                                // ... everyone's have, and everyone is done with plane p-1's buffer
        const f32x2 nk2 = -st_nxt;
        st_nxt = stat_load(p + 1);                                   // k2 of the next guide plane; behind plane 2: m
        if (p == 2) st_nx2 = stat_load(4);                           //                                             and 1 / sum
        // the centre pixel pair is read in front of the DMA issue (its LDS round trip passes in the shadow of the four pieces)
        f32x2 ctr;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ctr) : "v"(org + (p & 1) * (TR * TP * 4)), "n"((HALO * TP + HALO) * 4) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        stage(p + 1, (p + 1) & 1);                                   // (np >= 4) streams in behind the taps below
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ctr) :: "memory");
        if (EXCEL_DBG(dbg) & 1) continue;
        // guide channel p:  z_t += -(I_nb - I)^2 k2_p   as fma(dv dv, -k2, z), dv = nb + (-ctr): bit-identical to the affinity kernel.
        // (-ctr is opaque: the compiler otherwise folds the negation back and emits two unpacked v_sub_f32 instead of one v_pk_add_f32)
        f32x2 nctr = -ctr;
        if (EXCEL_DBG(dbg) & 16) nctr += 1e-30f * far_taps(guide + 3 * tg.base + (long long)p * HW);
        asm volatile("" : "+v"(nctr));
        taps(std::integral_constant<int, 1>{}, p & 1, [&](int di, int k, const f32x2 nb) {
            const f32x2 dv = nb + nctr;
            wall[di][k] = __builtin_elementwise_fma(dv * dv, nk2, wall[di][k]);
        }, [] {});
    }
    if (!(EXCEL_DBG(dbg) & 2)) {
        // aff_t = 2^(z_t - m) / sum + pos_t : the affinity kernel's operations, one rounding each (z + (-m) == z - m)
        f32x2 nm2 = -st_nxt;                                         // (plane 3's pieces had the taps of plane 2 to land)
        asm volatile("" : "+v"(nm2));
        const f32x2 is2 = st_nx2;
#pragma unroll
        for (int di = 0; di < ND; ++di)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f32x2 z = wall[di][k] + nm2;
                const f32x2 e = {__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
                const float ps = dl.pos_sm[di * 8 + k];
                wall[di][k] = __builtin_elementwise_fma(e, is2, f32x2{ps, ps});
            }
    }
    const bool valid = px < W && py < H;
    float* out_px = out + (long long)Cmax * tg.base + (long long)py * Wp + px;
    // behind the first mask plane the newest vmcnt event in front of a plane barrier is the previous plane's output store (one per wave,
    // if any of its lanes is inside the image): the pieces of plane p are older, so `vmcnt(1)` has them landed and leaves the store in
    // flight instead of waiting out its round trip at every plane
#ifndef PAR_STORE_WAIT
#define PAR_STORE_WAIT 1
#endif
    const bool wstore = PAR_STORE_WAIT && __builtin_amdgcn_ballot_w64(valid) != 0 && !EXCEL_DBG(dbg);
    for (int p = 3; p < np; ++p) {
        if (p > 3 && wstore) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (p + 1 < np) stage(p + 1, (p + 1) & 1);
        if (EXCEL_DBG(dbg) & 4) continue;
        f32x2 acc = {0.f, 0.f};
        if (EXCEL_DBG(dbg) & 16) acc = 1e-30f * far_taps(in + (long long)Cmax * tg.base + (long long)(p - 3) * HW);
        // (acc is pinned per tap row: it is only stored under `valid`, and the whole fma chain was otherwise sunk into that branch,
        //  behind all 50 reads of the plane)
        taps(std::integral_constant<int, PAR_MASK_DEPTH>{}, p & 1, [&](int di, int k, const f32x2 nb) { acc = __builtin_elementwise_fma(nb, wall[di][k], acc); },   // tap order, fused
             [&] { asm volatile("" : "+v"(acc)); });
        if (valid) *reinterpret_cast<f32x2*>(out_px + (long long)(p - 3) * HW) = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The COMPACT statistics (k2_r, k2_g, k2_b, m, 1/sum) of the 6-dilation set through the SAME tile machinery as the Jacobi step: a
// guide plane is staged once per 64 x 16 tile (LDS-DMA, halo 24) and its 48 taps are read three times from LDS - tap sum (-> mean),
// squared deviations (-> unbiased std -> k2), exponent accumulation - instead of 144 edge-clamped global loads per pixel and 48 live
// tap registers (par_affinity_kernel: 320 us per 32 x 448^2 step).  Same operations per pixel in the same order as
// par_affinity_kernel -> the same bits (tests: recompute == streamed, ragged == per-image).
template <bool RAGGED>
__global__ __launch_bounds__(512, 4) void par_stats_tile_kernel(const float* __restrict__ guide, float* __restrict__ stats, TileGeo geo, float w1) {
    constexpr int ND = 6, HALO = 24, TR = 16 + 2 * HALO, TP = PG_TP, NG = 3 * ND, NT = 8 * ND;
    __shared__ __attribute__((aligned(1024))) float tile[2 * TR * TP];   // [2][TR][TP]
    const Tile tg = tile_of<RAGGED>(geo);
    const int x0 = tg.x0, y0 = tg.y0, H = tg.H, W = tg.W, Wp = tg.Wp;
    const long long HW = tg.HW;
    const int tid = threadIdx.x, lane = tid & 63, tx = tid & 31, ty = tid >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = x0 + 2 * tx, py = y0 + ty;

    // ---- plane staging by LDS-DMA, as in par_iterate_guide_kernel
    typedef __attribute__((address_space(3))) unsigned char* lds_bptr;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(guide + 3 * tg.base), 0, (int)(3 * HW * 4), 0x00020000);
    const int xoffA = min(max(x0 - HALO + lane, 0), W - 1) * 4, xoffB = min(max(x0 - HALO + 64 + lane, 0), W - 1) * 4;   // replicate padding in x
    const unsigned tile_b = lds_addr(tile);
    constexpr int RPW = TR / 8;
    const bool interior = x0 - HALO >= 0 && x0 + 64 + HALO <= W;
    const int x4off = min(x0 - HALO + 4 * (lane & 31), Wp - 4) * 4;
    auto stage = [&](int p, int buf) {
        const int plane_off = p * (int)(HW * 4);
        unsigned dst = tile_b + (buf * TR + wave * RPW) * (TP * 4);
        asm volatile("" : "+s"(dst));
        if (interior) {
            int half = lane >> 5;
            asm volatile("" : "+v"(half));
#pragma unroll
            for (int j = 0; j < RPW / 2; ++j) {
                const int gy = min(max(y0 - HALO + wave * RPW + 2 * j + half, 0), H - 1);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_bptr)(unsigned long long)(dst + j * (2 * TP * 4)), 16, gy * Wp * 4 + x4off, plane_off, 0, 0);
            }
            return;
        }
#pragma unroll 4
        for (int rr = 0; rr < RPW; ++rr) {
            const int gy = min(max(y0 - HALO + wave * RPW + rr, 0), H - 1);
            const int soff = plane_off + gy * Wp * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_bptr)(unsigned long long)(dst + rr * (TP * 4)), 4, xoffA, soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_bptr)(unsigned long long)(dst + rr * (TP * 4) + 256), 4, xoffB, soff, 0, 0);
        }
    };

    f32x2 acc[ND][8];
#pragma unroll
    for (int di = 0; di < ND; ++di)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[di][k] = f32x2{0.f, 0.f};

    const unsigned org = tile_b + (ty * TP + 2 * tx) * 4;
    auto taps = [&](int buf, auto&& fn, auto&& pin) {            // (par_iterate_guide_kernel: tap row g+1 in flight behind row g)
        unsigned base = org + buf * (TR * TP * 4);
        asm volatile("" : "+v"(base));
        f32x2 sl[2][3];
        auto issue = [&](auto g) {
            constexpr int G = decltype(g)::value, r = G % 3, d = ParD<G / 3>::v, s = G & 1;
            constexpr int row = (HALO + (r - 1) * d) * TP * 4, cl = (d == 1) ? HALO - 2 : HALO - d, cr = (d == 1) ? HALO + 2 : HALO + d;
            sl[s][0] = lds_read8_imm<row + cl * 4>(base);
            if constexpr (r != 1 || d == 1) sl[s][1] = lds_read8_imm<row + HALO * 4>(base);
            sl[s][2] = lds_read8_imm<row + cr * 4>(base);
        };
        issue(std::integral_constant<int, 0>{});
        static_for<NG>([&](auto g) {
            constexpr int G = decltype(g)::value, di = G / 3, r = G % 3, d = ParD<di>::v, s = G & 1;
            constexpr bool mid = (r != 1 || d == 1);
            if constexpr (G + 1 < NG) {
                issue(std::integral_constant<int, G + 1>{});
                constexpr int nxt = ((G + 1) % 3 != 1 || ParD<(G + 1) / 3>::v == 1) ? 3 : 2;
                if constexpr (mid) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(sl[s][0]), "+v"(sl[s][1]), "+v"(sl[s][2]) : "n"(nxt) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(sl[s][0]), "+v"(sl[s][2]) : "n"(nxt) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sl[s][0]), "+v"(sl[s][1]), "+v"(sl[s][2])::"memory");
            }
            constexpr int k0 = (r == 0) ? 0 : (r == 1 ? 3 : 5);
            const f32x2 L = sl[s][0], R = sl[s][2];
            if constexpr (d == 1) {
                const f32x2 M = sl[s][1];
                fn(di, k0, f32x2{L[1], M[0]});
                if constexpr (r != 1) fn(di, k0 + 1, M);
                fn(di, k0 + (r != 1 ? 2 : 1), f32x2{M[1], R[0]});
            } else {
                fn(di, k0, L);
                if constexpr (r != 1) fn(di, k0 + 1, sl[s][1]);
                fn(di, k0 + (r != 1 ? 2 : 1), R);
            }
            pin();
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    const bool valid = px < W && py < H;
    float* st_px = stats + 5 * tg.base + (long long)py * Wp + px;
    stage(0, 0);
#pragma unroll 1
    for (int p = 0; p < 3; ++p) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (p + 1 < 3) stage(p + 1, (p + 1) & 1);
        f32x2 ctr;
        asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(ctr) : "v"(org + (p & 1) * (TR * TP * 4)), "n"((HALO * TP + HALO) * 4) : "memory");
        // unbiased std over the 48 taps (torch.std): tap sum in tap order, then the squared deviations from the mean in tap order
        f32x2 sum = {0.f, 0.f};
        taps(p & 1, [&](int, int, const f32x2 nb) { sum += nb; }, [&] { asm volatile("" : "+v"(sum)); });
        const f32x2 mean = {sum[0] / (float)NT, sum[1] / (float)NT};
        f32x2 nmean = -mean;
        asm volatile("" : "+v"(nmean));
        f32x2 var = {0.f, 0.f};
        taps(p & 1, [&](int, int, const f32x2 nb) { const f32x2 dv = nb + nmean; var = __builtin_elementwise_fma(dv, dv, var); },
             [&] { asm volatile("" : "+v"(var)); });
        f32x2 k2;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float den = sqrtf(var[q] / (float)(NT - 1)) + 1e-8f;
            const float k = 1.f / (den * w1);
            k2[q] = __fmul_rn(k * k, PAR_LOG2E_3);
        }
        if (valid) *reinterpret_cast<f32x2*>(st_px + (long long)p * HW) = k2;
        f32x2 nk2 = -k2, nctr = -ctr;
        asm volatile("" : "+v"(nctr));
        taps(p & 1, [&](int di, int k, const f32x2 nb) {
            const f32x2 dv = nb + nctr;
            acc[di][k] = __builtin_elementwise_fma(dv * dv, nk2, acc[di][k]);
        }, [] {});
    }
    f32x2 m = {-INFINITY, -INFINITY};
#pragma unroll
    for (int di = 0; di < ND; ++di)
#pragma unroll
        for (int k = 0; k < 8; ++k) { m[0] = fmaxf(m[0], acc[di][k][0]); m[1] = fmaxf(m[1], acc[di][k][1]); }
    f32x2 nm = -m;
    asm volatile("" : "+v"(nm));
    f32x2 ssum = {0.f, 0.f};
#pragma unroll
    for (int di = 0; di < ND; ++di)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x2 z = acc[di][k] + nm;
            ssum += f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
        }
    if (valid) {
        *reinterpret_cast<f32x2*>(st_px + 3 * HW) = m;
        *reinterpret_cast<f32x2*>(st_px + 4 * HW) = f32x2{1.f / ssum[0], 1.f / ssum[1]};
    }
}

// F.interpolate(mode='bilinear', align_corners=True) (PAR.py:67): planes of h x w -> H x W
__device__ __forceinline__ float bilinear_ac_px(const float* __restrict__ p, int h, int w, int H, int W, int x, int y) {
    const float sy = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float top = fmaf(lx, p[y0 * w + x1], __fmul_rn(1.f - lx, p[y0 * w + x0]));
    const float bot = fmaf(lx, p[y1 * w + x1], __fmul_rn(1.f - lx, p[y1 * w + x0]));
    return fmaf(ly, bot, __fmul_rn(1.f - ly, top));
}

__global__ __launch_bounds__(256) void bilinear_ac_kernel(const float* __restrict__ in, float* __restrict__ out, int planes,
                                                          int h, int w, int H, int W) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)planes * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long pl = i / ((long long)W * H);
    out[i] = bilinear_ac_px(in + pl * h * w, h, w, H, W, x, y);
}

// ragged: in [B,3,h,w] (uniform) -> the pitched guide planes of every image at its own (H_b, W_b); grid (tiles, 3 channels)
__global__ __launch_bounds__(256) void bilinear_ac_ragged_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w, TileGeo geo) {
    const Tile t = tile_of<true>(geo);
    const int c = blockIdx.y;
    const int x = t.x0 + (threadIdx.x & 63);
    if (x >= t.W) return;
    const float* p = in + ((long long)t.b * 3 + c) * h * w;
    float* o = out + 3 * t.base + c * t.HW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = t.y0 + (threadIdx.x >> 6) + 4 * r;
        if (y < t.H) o[(long long)y * t.Wp + x] = bilinear_ac_px(p, h, w, t.H, t.W, x, y);
    }
}

__device__ __forceinline__ int argmax_key(const float* __restrict__ p, long long HW, int nch, const int* __restrict__ cls_row) {
    float best = p[0];
    int bi = 0;
    for (int c = 1; c < nch; ++c) {
        const float v = p[(long long)c * HW];
        if (v > best) { best = v; bi = c; }       // first maximum wins, like torch.argmax
    }
    // valid_key = [0, cls+1 ...] (affutils.py:168)
    return (bi == 0) ? 0 : (cls_row ? cls_row[bi - 1] + 1 : bi);
}

__global__ __launch_bounds__(256) void argmax_label_kernel(const float* __restrict__ cams, const int* __restrict__ nchan,
                                                           const int* __restrict__ cls_idx, int Smax, int Cmax,
                                                           long long HW, unsigned char* __restrict__ lab8,
                                                           long long* __restrict__ lab64) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const int nch = nchan ? min(nchan[b], Cmax) : Cmax;
    const int key = argmax_key(cams + (long long)b * Cmax * HW + i, HW, nch, cls_idx ? cls_idx + (long long)b * Smax : nullptr);
    if (lab8) lab8[(long long)b * HW + i] = (unsigned char)key;
    if (lab64) lab64[(long long)b * HW + i] = key;
}

// ragged: pitched cams -> TIGHT u8 labels (image b at loff_b, row pitch W_b): the confusion kernel then runs over one flat array
__global__ __launch_bounds__(256) void argmax_label_ragged_kernel(const float* __restrict__ cams, const int* __restrict__ nchan,
                                                                  const int* __restrict__ cls_idx, int Smax, int Cmax, TileGeo geo,
                                                                  unsigned char* __restrict__ lab8) {
    const Tile t = tile_of<true>(geo);
    const int x = t.x0 + (threadIdx.x & 63);
    if (x >= t.W) return;
    const int nch = nchan ? min(nchan[t.b], Cmax) : Cmax;
    const int* cls_row = cls_idx ? cls_idx + (long long)t.b * Smax : nullptr;
    const float* cb = cams + (long long)Cmax * t.base;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = t.y0 + (threadIdx.x >> 6) + 4 * r;
        if (y < t.H) lab8[t.lab + (long long)y * t.W + x] = (unsigned char)argmax_key(cb + (long long)y * t.Wp + x, t.HW, nch, cls_row);
    }
}

#define CONF_MAXBINS 8192
__global__ __launch_bounds__(256) void confusion_kernel(const unsigned char* __restrict__ gt, const unsigned char* __restrict__ pred,
                                                        long long n, int nc, unsigned long long* __restrict__ hist, int head) {
    __shared__ unsigned int lh[CONF_MAXBINS];
    const int bins = nc * nc;
    for (int i = threadIdx.x; i < bins; i += 256) lh[i] = 0;
    __syncthreads();
    // gt / pred slices of a larger tensor need not be 16-byte aligned: when both share the same misalignment a scalar head brings
    // them to a 16-byte boundary, otherwise (`head` < 0) the whole range goes through the scalar path
    if (head > 0 && blockIdx.x == 0 && threadIdx.x < head && threadIdx.x < n) {
        const int g = gt[threadIdx.x], p = pred[threadIdx.x];
        if (g < nc && p < nc) atomicAdd(&lh[g * nc + p], 1u);
    }
    if (head >= 0) {
        gt += head; pred += head; n -= head;
        if (n < 0) n = 0;
    }
    const long long stride = (long long)gridDim.x * 256 * 16;
    for (long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 16; base < n; base += stride) {
        if (head >= 0 && base + 16 <= n) {
            const uint4 g4 = *reinterpret_cast<const uint4*>(gt + base);
            const uint4 p4 = *reinterpret_cast<const uint4*>(pred + base);
            const unsigned int gw[4] = {g4.x, g4.y, g4.z, g4.w}, pw[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int g = (gw[k >> 2] >> (8 * (k & 3))) & 255, p = (pw[k >> 2] >> (8 * (k & 3))) & 255;
                if (g < nc && p < nc) atomicAdd(&lh[g * nc + p], 1u);
            }
        } else {
            const long long end = (base + 16 < n) ? base + 16 : n;
            for (long long j = base; j < end; ++j) {
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
    for (int i = 0; i < 8; ++i) out->d[i] = 0;
    for (int i = 0; i < ndil; ++i) {
        EXCEL_CHECK_ARG(dil[i] >= 1, "PAR: dilations must be >= 1 (got %d)", dil[i]);
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

// The recomputing step is built for the dilation set every caller of the reference uses, [1,2,4,8,12,24] (tools/infer_lam.py:168,
// scripts/train_voc.py:112, scripts/train_coco.py:110): 6 x 8 float2 weights stay in registers (96 of 128 VGPRs), halo 24, tap
// offsets as compile-time immediates.  Any other set goes through the streamed kernel.
static bool par_dil_ok(const int* dil, int ndil) {
    static const int want[6] = {1, 2, 4, 8, 12, 24};
    if (ndil != 6) return false;
    for (int i = 0; i < 6; ++i)
        if (dil[i] != want[i]) return false;
    return true;
}
// geo.tab != nullptr: ragged batch of `total_tiles` 64x16 tiles (img / aff in the pitched layout); else uniform [B,.,H,W]
int excel_launch_par_affinity(const float* img, float* aff, const TileGeo& geo, int total_tiles, const int* dil, int ndil, float w1, float w2,
                              hipStream_t st, int compact) {
    ProfScope prof__(PROF_PAR_AFFINITY, st);
    ParDil dl;
    int rc = make_dil(dil, ndil, w1, w2, &dl);
    if (rc) return rc;
    // compact statistics of the standard dilation set, 16-byte aligned pitched planes: the tile kernel
    const bool tiled = compact && par_dil_ok(dil, ndil) && ((((uintptr_t)img | (uintptr_t)aff) & 15) == 0) &&
                       (geo.tab ? true : (geo.W % 4 == 0 && geo.W >= 8 && 5LL * geo.H * geo.W * 4 < (1LL << 31)));
    if (geo.tab) {
        EXCEL_CHECK_ARG(compact && ndil == 6, "par_affinity: ragged batches use the compact statistics of the 6-dilation kernel");
        if (tiled) hipLaunchKernelGGL((par_stats_tile_kernel<true>), dim3(total_tiles), dim3(512), 0, st, img, aff, geo, w1);
        else hipLaunchKernelGGL((par_affinity_kernel<6, true>), dim3(total_tiles, 4), dim3(256), 0, st, img, aff, dl, geo, w1, 1);
    } else if (tiled) {
        hipLaunchKernelGGL((par_stats_tile_kernel<false>), dim3(cdiv(geo.W, 64), cdiv(geo.H, 16), geo.B), dim3(512), 0, st, img, aff, geo, w1);
    } else {
#define CALL(N) hipLaunchKernelGGL((par_affinity_kernel<N, false>), dim3(cdiv(geo.W, 64), cdiv(geo.H, 4), geo.B), dim3(256), 0, st, img, aff, dl, geo, w1, compact ? 1 : 0)
        ND_SWITCH(ndil, CALL)
#undef CALL
    }
    EXCEL_CHECK_LAUNCH("par_affinity");
    return EXCEL_OK;
}

int excel_launch_par_iterate(const float* aff, const float* in, float* out, const int* nchan, int B, int Cmax, int H, int W,
                             const int* dil, int ndil, hipStream_t st) {
    ProfScope prof__(PROF_PAR_ITERATE, st);
    ParDil dl;
    int rc = make_dil(dil, ndil, 0.3f, 0.01f, &dl);
    if (rc) return rc;
    hipLaunchKernelGGL(par_iterate_stream_kernel, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(256), 0, st, aff, in, out, nchan, dl, ndil, Cmax, H, W);
    EXCEL_CHECK_LAUNCH("par_iterate");
    return EXCEL_OK;
}

// max_plane: the largest H * Wp of the launch; the LDS-DMA offsets inside one image are 32-bit (buffer descriptor per image)
int excel_par_guide_supported(const void* guide, const void* stats, const void* in, const void* out, int Cmax, long long max_plane, int Wp,
                              const int* dil, int ndil) {
    // (any pitch that is a multiple of 4 floats works, down to Wp = 4: border tiles clamp their source columns per lane -
    // tests/test_gpu_ops.py::test_ragged_tiny_images, test_par_uniform_narrow_images)
    const bool vec = (Wp % 4) == 0 && Wp >= 4 && ((((uintptr_t)guide | (uintptr_t)stats | (uintptr_t)in | (uintptr_t)out) & 15) == 0);
    const long long planes = Cmax > 5 ? Cmax : 5;
    return (vec && par_dil_ok(dil, ndil) && planes * max_plane * 4 < (1LL << 31)) ? 1 : 0;
}

// One Jacobi step with the affinities recomputed from (guide, stats).  The caller has checked excel_par_guide_supported.
int excel_launch_par_iterate_guide(const float* guide, const float* stats, const float* in, float* out, const int* nchan, int Cmax,
                                   const TileGeo& geo, int total_tiles, const int* dil, int ndil, float w1, float w2, hipStream_t st) {
    EXCEL_CHECK_ARG(par_dil_ok(dil, ndil), "par_iterate_guide: unsupported dilation set");
    ProfScope prof__(PROF_PAR_ITERATE, st);
    ParDil dl;
    int rc = make_dil(dil, ndil, w1, w2, &dl);
    if (rc) return rc;
    int dbg = 0;
#ifdef EXCEL_DEV
    static const int env_dbg = getenv("EXCEL_PAR_DBG") ? atoi(getenv("EXCEL_PAR_DBG")) : 0;
    dbg = env_dbg;
#endif
    if (geo.tab)
        hipLaunchKernelGGL((par_iterate_guide_kernel<true>), dim3(total_tiles), dim3(512), 0, st, guide, stats, in, out, nchan, dl, Cmax, geo, dbg);
    else
        hipLaunchKernelGGL((par_iterate_guide_kernel<false>), dim3(cdiv(geo.W, 64), cdiv(geo.H, 16), geo.B), dim3(512), 0, st, guide, stats,
                           in, out, nchan, dl, Cmax, geo, dbg);
    EXCEL_CHECK_LAUNCH("par_iterate_guide");
    return EXCEL_OK;
}

int excel_launch_bilinear_ac(const float* in, float* out, int planes, int h, int w, int H, int W, hipStream_t st) {
    ProfScope prof__(PROF_OTHER, st);
    const long long n = (long long)planes * H * W;
    hipLaunchKernelGGL(bilinear_ac_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, in, out, planes, h, w, H, W);
    EXCEL_CHECK_LAUNCH("bilinear_ac");
    return EXCEL_OK;
}

int excel_launch_bilinear_ac_ragged(const float* in, float* out, int h, int w, const TileGeo& geo, int total_tiles, hipStream_t st) {
    ProfScope prof__(PROF_OTHER, st);
    hipLaunchKernelGGL(bilinear_ac_ragged_kernel, dim3(total_tiles, 3), dim3(256), 0, st, in, out, h, w, geo);
    EXCEL_CHECK_LAUNCH("bilinear_ac_ragged");
    return EXCEL_OK;
}

int excel_launch_argmax_label(const float* cams, const int* nchan, const int* cls_idx, int B, int Smax, int Cmax, long long HW,
                              unsigned char* lab8, long long* lab64, hipStream_t st) {
    ProfScope prof__(PROF_ARGMAX, st);
    hipLaunchKernelGGL(argmax_label_kernel, dim3((unsigned)cdivl(HW, 256), B), dim3(256), 0, st, cams, nchan, cls_idx, Smax, Cmax, HW, lab8, lab64);
    EXCEL_CHECK_LAUNCH("argmax_label");
    return EXCEL_OK;
}

int excel_launch_argmax_label_ragged(const float* cams, const int* nchan, const int* cls_idx, int Smax, int Cmax, const TileGeo& geo,
                                     int total_tiles, unsigned char* lab8, hipStream_t st) {
    ProfScope prof__(PROF_ARGMAX, st);
    hipLaunchKernelGGL(argmax_label_ragged_kernel, dim3(total_tiles), dim3(256), 0, st, cams, nchan, cls_idx, Smax, Cmax, geo, lab8);
    EXCEL_CHECK_LAUNCH("argmax_label_ragged");
    return EXCEL_OK;
}

int excel_launch_confusion(const unsigned char* gt, const unsigned char* pred, long long n, int nc, unsigned long long* hist,
                           hipStream_t st) {
    ProfScope prof__(PROF_CONFUSION, st);
    EXCEL_CHECK_ARG(nc >= 1 && nc * nc <= CONF_MAXBINS, "confusion: num_classes %d too large", nc);
    // scalar elements in front of the first 16-byte boundary (same for both pointers), or -1: no common alignment -> scalar path
    const int mg = (int)((16 - ((uintptr_t)gt & 15)) & 15), mp = (int)((16 - ((uintptr_t)pred & 15)) & 15);
    const int head = (mg == mp) ? mg : -1;
    const int blocks = (int)((cdivl(n, 256 * 16) < 2048) ? cdivl(n, 256 * 16) : 2048);
    hipLaunchKernelGGL(confusion_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, gt, pred, n, nc, hist, head);
    EXCEL_CHECK_LAUNCH("confusion");
    return EXCEL_OK;
}
