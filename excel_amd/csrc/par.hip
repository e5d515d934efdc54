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
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "excel_internal.h"

struct ParDil {
    int d[8];
    float pos_sm[64];   // w2 * softmax over taps of the position term (constant vector, PAR.py:83,86)
};

__device__ __constant__ int TAP_DY[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
__device__ __constant__ int TAP_DX[8] = {-1, 0, 1, -1, 1, -1, 0, 1};

// COMPACT: instead of the 48 affinity planes write the 5 per-pixel statistics they are a function of,
//   stats[b][0..2] = k2_c = 1 / ((std_c + 1e-8) w1)^2,   stats[b][3] = m (max of the 48 exponents),   stats[b][4] = 1 / sum exp
// par_iterate_guide_kernel recomputes  aff_t = exp(sum_c -(I_nb - I)^2 k2_c / 3 - m) / sum + pos_t  from them with the SAME operations
// in the same order, i.e. bit-identical weights, from 20 B/pixel instead of 192 B/pixel.
// (a RUN-TIME flag on one instantiation: both output forms come from the same compiled arithmetic.)
template <int ND>
__global__ __launch_bounds__(256) void par_affinity_kernel(const float* __restrict__ img, float* __restrict__ aff, ParDil dl,
                                                           int H, int W, float w1, int COMPACT) {
    constexpr int NT = 8 * ND;
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    // one wave = one image row: y and every tap row are wave-uniform -> scalar row bases, per-lane work is the x offset only
    // (per-lane 64-bit address arithmetic for the 144 tap loads was most of this kernel's VALU work)
    const int y = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (y >= H) return;
    const bool in_x = x < W;
    const int xc = min(x, W - 1);
    const long long HW = (long long)H * W;
    int xo[ND][2];
#pragma unroll
    for (int di = 0; di < ND; ++di) { xo[di][0] = max(xc - dl.d[di], 0); xo[di][1] = min(xc + dl.d[di], W - 1); }
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const float* ch = img + ((long long)b * 3 + c) * HW;
        const float* row0 = ch + (long long)y * W;
        const float ctr = row0[xc];
        float nb[NT];
        float sum = 0.f;
#pragma unroll
        for (int di = 0; di < ND; ++di) {
            const int d = dl.d[di];
            const float* rup = ch + (long long)max(y - d, 0) * W;          // scalar
            const float* rdn = ch + (long long)min(y + d, H - 1) * W;
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
        for (int t = 0; t < NT; ++t) { const float dv = nb[t] - mean; var += dv * dv; }
        const float den = sqrtf(var / (float)(NT - 1)) + 1e-8f;     // unbiased std (torch.std default)
        // (|nb - ctr| / den / w1)^2 as one fma per tap: 3 true divisions per pixel instead of 2 x 144 (the divisions
        // were ~80 % of this kernel's instructions); differs from the literal form by one rounding (~1e-7 relative)
        const float k = 1.f / (den * w1);
        const float k2 = k * k;
        if (COMPACT && in_x) aff[((long long)b * 5 + c) * HW + (long long)y * W + x] = k2;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float dv = nb[t] - ctr;
            acc[t] = fmaf(-(dv * dv), k2, acc[t]);
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t] = __fmul_rn(acc[t], 1.f / 3.f); m = fmaxf(m, acc[t]); }     // mean over RGB (no contraction:
    float s = 0.f;                                                                                   //  par_iterate_guide_kernel redoes it)
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t] = __expf(__fsub_rn(acc[t], m)); s += acc[t]; }
    const float inv_s = 1.f / s;
    if (!in_x) return;
    if (COMPACT) {
        aff[((long long)b * 5 + 3) * HW + (long long)y * W + x] = m;
        aff[((long long)b * 5 + 4) * HW + (long long)y * W + x] = inv_s;
        return;
    }
    float* out = aff + (long long)b * NT * HW + (long long)y * W + x;
#pragma unroll
    for (int t = 0; t < NT; ++t) out[(long long)t * HW] = fmaf(acc[t], inv_s, dl.pos_sm[t]);
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

// ---------------------------------------------------------------------------------------------------------------
// Vectorised Jacobi step: one thread = 4 consecutive pixels of a row, every global access a 16-byte float4:
//   * the aff stream (the dominant traffic, 48 floats/pixel) is read with full-width coalesced float4 loads;
//   * taps whose dilation is a multiple of 4 (4, 8, 12, 24) are aligned float4 loads of the shifted group;
//   * dilations 1..3 read the 12-float window [x0-4, x0+8) of the row once (3 float4) and pick the shifted values
//     from registers;
//   * replicate padding: rows clamp by index; column groups are either fully inside or fully outside (W % 4 == 0),
//     an outside group is a splat of the edge pixel -> branch-free, no per-element clamps.
//   * channel count is a compile-time constant per image (switch on the block-uniform nchan[b]): no predicated
//     loads, so hipcc keeps every load in flight instead of serialising round trips.
__device__ __forceinline__ f32x4 par_ldg4(const float* __restrict__ rowp, int xg, int W) {
    const int xc = min(max(xg, 0), W - 4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + xc);
    // branch-free edge replication (selects, so the load is never wrapped in an exec-masked branch)
    const bool lo = xg < 0, out = lo || (xg > W - 4);
    const float e = lo ? v[0] : v[3];
    return f32x4{out ? e : v[0], out ? e : v[1], out ? e : v[2], out ? e : v[3]};
}

template <int ND, int NC>
__device__ __forceinline__ void par_body4(const float* __restrict__ a, const float* __restrict__ ib, float* __restrict__ ob,
                                          const int (&dils)[8], int x0, int y, int H, int W, long long HW) {
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int di = 0; di < ND; ++di) {   // not unrolled: 8 aff float4 + one dilation's taps in flight keeps VGPRs low
        int d = dils[0];                // select chain with static indices: a dynamically indexed kernarg struct goes to scratch
#pragma unroll
        for (int q = 1; q < ND; ++q) d = (di == q) ? dils[q] : d;
        f32x4 w8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w8[k] = *reinterpret_cast<const f32x4*>(a + (long long)(di * 8 + k) * HW);
        const int yy[3] = {min(max(y - d, 0), H - 1), y, min(max(y + d, 0), H - 1)};
        if ((d & 3) == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float* ch = ib + (long long)c * HW;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float* rowp = ch + (long long)yy[r] * W;
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        if (r == 1 && dx == 0) continue;
                        const int k = (r == 0) ? dx + 1 : (r == 1 ? (dx < 0 ? 3 : 4) : dx + 6);
                        acc[c] += par_ldg4(rowp, x0 + dx * d, W) * w8[k];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // keep one channel's 8 loads in flight, not NC x 8 (VGPR cap)
            }
        } else {   // 1 <= d <= 3: register window; d is wave-uniform -> scalar branch over three fixed shifts
            auto window = [&](auto shift_tag) {
                constexpr int SH = decltype(shift_tag)::value;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float* ch = ib + (long long)c * HW;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float* rowp = ch + (long long)yy[r] * W;
                        const f32x4 L = par_ldg4(rowp, x0 - 4, W), M = par_ldg4(rowp, x0, W), R = par_ldg4(rowp, x0 + 4, W);
                        const float win[12] = {L[0], L[1], L[2], L[3], M[0], M[1], M[2], M[3], R[0], R[1], R[2], R[3]};
#pragma unroll
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (r == 1 && dx == 0) continue;
                            const int k = (r == 0) ? dx + 1 : (r == 1 ? (dx < 0 ? 3 : 4) : dx + 6);
                            const f32x4 v = {win[4 + SH * dx], win[5 + SH * dx], win[6 + SH * dx], win[7 + SH * dx]};
                            acc[c] += v * w8[k];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (d == 1) window(std::integral_constant<int, 1>{});
            else if (d == 2) window(std::integral_constant<int, 2>{});
            else window(std::integral_constant<int, 3>{});
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) *reinterpret_cast<f32x4*>(ob + (long long)c * HW) = acc[c];
}

template <int ND>
__global__ __launch_bounds__(256) void par_iterate4_kernel(const float* __restrict__ aff, const float* __restrict__ in,
                                                           float* __restrict__ out, const int* __restrict__ nchan, ParDil dl,
                                                           int Cmax, int H, int W) {
    constexpr int NT = 8 * ND;
    const int b = blockIdx.z;
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x0 >= W || y >= H) return;
    const int nch = nchan ? min(nchan[b], Cmax) : Cmax;
    const long long HW = (long long)H * W;
    const float* a = aff + (long long)b * NT * HW + (long long)y * W + x0;
    const float* ib = in + (long long)b * Cmax * HW;
    float* ob = out + (long long)b * Cmax * HW + (long long)y * W + x0;
    const int dils[8] = {dl.d[0], dl.d[1], dl.d[2], dl.d[3], dl.d[4], dl.d[5], dl.d[6], dl.d[7]};
    for (int c0 = 0; c0 < nch; c0 += 4) {
        const float* ic = ib + (long long)c0 * HW;
        float* oc = ob + (long long)c0 * HW;
        switch (min(nch - c0, 4)) {
            case 1: par_body4<ND, 1>(a, ic, oc, dils, x0, y, H, W, HW); break;
            case 2: par_body4<ND, 2>(a, ic, oc, dils, x0, y, H, W, HW); break;
            case 3: par_body4<ND, 3>(a, ic, oc, dils, x0, y, H, W, HW); break;
            default: par_body4<ND, 4>(a, ic, oc, dils, x0, y, H, W, HW); break;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-tiled Jacobi step (the production kernel).  One workgroup = a 64 x 16 pixel tile, one thread = 4 pixels.
//   1. every thread issues its 48 aff float4 loads up front (the HBM stream: 192 B/pixel, fully coalesced,
//      all in flight at once -> bandwidth-, not latency-bound) and keeps them in registers;
//   2. the mask tile + replicate-padded halo (max dilation, here 24 -> 64 x 112 floats per channel) is staged
//      through LDS two channels at a time with aligned float4 loads (L2 hits: masks are 0.8 MB/channel/image);
//   3. taps are conflict-free ds_read_b128: dilations that are multiples of 4 read the shifted aligned group,
//      dilations 1..3 read the 3 aligned groups around the pixel once per row and shift in registers.
// aff is read exactly once per step whatever the channel count (it stays in registers across channel pairs).
// One channel pair of a tile: stage the pair's mask tile (+ halo) in LDS, accumulate the 8*ND taps.
// WSRC(di, k) yields the aff float4 of tap (di, k): either the rolling double buffer (loads issued here, one dilation
// ahead) or the per-thread register copy of all taps (KEEP: images with more than two channels read aff ONCE for all
// their channel pairs instead of once per pair -- 41 % of VOC images have 3+ channels).
template <int ND, int HALO, int KEEPN>
__device__ __forceinline__ void par_lds_pair(const float* __restrict__ ac, const f32x4 (&wall)[KEEPN ? KEEPN : 1][8], float* tile,
                                             const float* __restrict__ in_pair, float* __restrict__ out_px, int nc, const ParDil& dl,
                                             int x0, int y0, int tid, int tx, int ty, bool valid, int H, int W, long long HW) {
    constexpr int halo = HALO, TR = 16 + 2 * HALO, TP = 64 + 2 * HALO, TP4 = TP >> 2;
    // opaque per call: otherwise LICM hoists the ~50 LDS tap offsets out of the caller's channel-pair loop and they
    // compete with the pinned aff registers (spills)
    asm volatile("" : "+v"(tx), "+v"(ty));
    const int cb = halo + 4 * tx;
    constexpr bool KEEP = KEEPN > 0;
    f32x4 wbuf[2][8];
    if (KEEPN < ND) {
        // the first streamed dilation's loads are issued BEFORE the tile staging so their HBM latency overlaps it
#pragma unroll
        for (int k = 0; k < 8; ++k) wbuf[KEEPN & 1][k] = *reinterpret_cast<const f32x4*>(ac + (long long)(KEEPN * 8 + k) * HW);
    }
    __syncthreads();
    const int per_ch = TR * TP4;
    for (int i = tid; i < nc * per_ch; i += 256) {
        const int ch = i >= per_ch ? 1 : 0;
        const int rem = i - ch * per_ch;
        const int r = rem / TP4, g = rem - r * TP4;
        const int gy = min(max(y0 - halo + r, 0), H - 1);
        const f32x4 v = par_ldg4(in_pair + (long long)ch * HW + (long long)gy * W, x0 - halo + 4 * g, W);
        *reinterpret_cast<f32x4*>(&tile[(ch * TR + r) * TP + 4 * g]) = v;
    }
    __syncthreads();
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int di = 0; di < ND; ++di) {
        if (di >= KEEPN && di + 1 < ND) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                wbuf[(di + 1) & 1][k] = *reinterpret_cast<const f32x4*>(ac + (long long)((di + 1) * 8 + k) * HW);
        }
        const f32x4 (&w8)[8] = di < KEEPN ? wall[di < KEEPN ? di : 0] : wbuf[di & 1];
        const int d = dl.d[di];
        const int rr[3] = {ty + halo - d, ty + halo, ty + halo + d};
        if ((d & 3) == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if (r == 1 && dx == 0) continue;
                    const int k = (r == 0) ? dx + 1 : (r == 1 ? (dx < 0 ? 3 : 4) : dx + 6);
                    const int off = rr[r] * TP + cb + dx * d;
                    // explicit fused multiply-adds in tap order: the result does not depend on how an instantiation is contracted
                    acc0 = __builtin_elementwise_fma(*reinterpret_cast<const f32x4*>(&tile[off]), w8[k], acc0);
                    if (nc > 1) acc1 = __builtin_elementwise_fma(*reinterpret_cast<const f32x4*>(&tile[TR * TP + off]), w8[k], acc1);
                    if (KEEP && dx == 1) __builtin_amdgcn_sched_barrier(0);
                }
        } else {
            auto window = [&](auto shift_tag) {
                constexpr int SH = decltype(shift_tag)::value;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    if (ch == 1 && nc < 2) break;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float* rowp = &tile[(ch * TR + rr[r]) * TP + cb];
                        const f32x4 L = *reinterpret_cast<const f32x4*>(rowp - 4), M = *reinterpret_cast<const f32x4*>(rowp),
                                    R = *reinterpret_cast<const f32x4*>(rowp + 4);
                        const float win[12] = {L[0], L[1], L[2], L[3], M[0], M[1], M[2], M[3], R[0], R[1], R[2], R[3]};
#pragma unroll
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (r == 1 && dx == 0) continue;
                            const int k = (r == 0) ? dx + 1 : (r == 1 ? (dx < 0 ? 3 : 4) : dx + 6);
                            const f32x4 v = {win[4 + SH * dx], win[5 + SH * dx], win[6 + SH * dx], win[7 + SH * dx]};
                            if (ch == 0) acc0 = __builtin_elementwise_fma(v, w8[k], acc0); else acc1 = __builtin_elementwise_fma(v, w8[k], acc1);
                        }
                        if (KEEP) __builtin_amdgcn_sched_barrier(0);      // one row's window live at a time (192 aff VGPRs are pinned)
                    }
                }
            };
            if (d == 1) window(std::integral_constant<int, 1>{});
            else if (d == 2) window(std::integral_constant<int, 2>{});
            else window(std::integral_constant<int, 3>{});
        }
        __builtin_amdgcn_sched_barrier(0);   // do not hoist later dilations' loads (aff or LDS) above this point (VGPR cap)
    }
    if (valid) {
        *reinterpret_cast<f32x4*>(out_px) = acc0;
        if (nc > 1) *reinterpret_cast<f32x4*>(out_px + HW) = acc1;
    }
}

template <int ND, int HALO>
__global__ __launch_bounds__(256, 2) void par_iterate_lds_kernel(const float* __restrict__ aff, const float* __restrict__ in,
                                                                 float* __restrict__ out, const int* __restrict__ nchan,
                                                                 ParDil dl, int Cmax, int H, int W) {
    constexpr int NT = 8 * ND;
    constexpr int TR = 16 + 2 * HALO, TP = 64 + 2 * HALO;
    __shared__ __attribute__((aligned(16))) float tile[2 * TR * TP];   // [2][TR][TP]
    const int b = blockIdx.z, x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int px = x0 + 4 * tx, py = y0 + ty;
    const bool valid = px < W && py < H;
    const int nch = nchan ? min(nchan[b], Cmax) : Cmax;
    const long long HW = (long long)H * W;
    const float* a = aff + (long long)b * NT * HW + (long long)min(py, H - 1) * W + min(px, W - 4);
    const float* in_b = in + (long long)b * Cmax * HW;
    float* out_px = out + (long long)b * Cmax * HW + (long long)py * W + px;
    if (nch > 2) {
        // two or more channel pairs: the workgroup is LDS-limited to 2 waves/SIMD anyway (256 VGPRs available), so the
        // aff float4 of this thread's 4 pixels stay in registers across the pairs
        constexpr int KEEPN = ND;
        f32x4 wall[KEEPN][8];
#pragma unroll
        for (int di = 0; di < KEEPN; ++di)
#pragma unroll
            for (int k = 0; k < 8; ++k) wall[di][k] = *reinterpret_cast<const f32x4*>(a + (long long)(di * 8 + k) * HW);
        for (int c0 = 0; c0 < nch; c0 += 2)
            par_lds_pair<ND, HALO, KEEPN>(a, wall, tile, in_b + (long long)c0 * HW, out_px + (long long)c0 * HW, min(2, nch - c0), dl,
                                         x0, y0, tid, tx, ty, valid, H, W, HW);
    } else {
        f32x4 none[1][8];
        par_lds_pair<ND, HALO, 0>(a, none, tile, in_b, out_px, nch, dl, x0, y0, tid, tx, ty, valid, H, W, HW);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Jacobi step that RECOMPUTES the affinities from the guide image (par_affinity_kernel COMPACT statistics).
// The 48 weights of a pixel are a function of 3 image values per tap and 5 per-pixel statistics; streaming them as 48 fp32
// planes made the step HBM-bound at (48 + 2C) x 4 B/pixel - here a step reads (5 + 3 + 2C) x 4 B/pixel and pays
// ~10 VALU operations per tap.  One workgroup = a 64 x 16 pixel tile (thread = 4 pixels); the planes it needs - the three guide
// channels, then the image's mask channels - go through a DOUBLE-BUFFERED LDS tile (64 rows x 128 floats incl. the 24-pixel halo),
// filled by 4-byte LDS-DMA with edge-clamped per-lane source addresses (= replicate padding): plane p+1 streams in while the taps
// of plane p are evaluated, one barrier per plane, no staging registers.  Phase 1 (guide planes) accumulates the tap exponents in
// wall[ND][8] (float4 = this thread's 4 pixels), a finalisation turns them into the weights, phase 2 (mask planes) applies them.
// Same operations in the same order as par_affinity_kernel + the streamed-plane kernel -> bit-identical results.
#define PG_TP 128            // LDS row pitch (floats): 512 B keeps every ds_read_b128 group on distinct banks
template <int ND, int HALO>
__global__ __launch_bounds__(256, 2) void par_iterate_guide_kernel(const float* __restrict__ guide, const float* __restrict__ stats,
                                                                   const float* __restrict__ in, float* __restrict__ out,
                                                                   const int* __restrict__ nchan, ParDil dl, int Cmax, int H, int W, int dbg) {
    constexpr int TR = 16 + 2 * HALO, TP = PG_TP;
    static_assert(64 + 2 * HALO <= TP, "tile row does not fit the LDS pitch");
    __shared__ __attribute__((aligned(1024))) float tile[2 * TR * TP];   // [2][TR][TP]
    const int b = blockIdx.z, x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, tx = tid & 15, ty = tid >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = x0 + 4 * tx, py = y0 + ty;
    const bool valid = px < W && py < H;
    const int nch = nchan ? min(nchan[b], Cmax) : Cmax;
    const long long HW = (long long)H * W;
    const long long pix = (long long)min(py, H - 1) * W + min(px, W - 4);
    const float* st_b = stats + (long long)b * 5 * HW + pix;

    // ---- plane staging by LDS-DMA: wave w fills rows [w*TR/4, (w+1)*TR/4) of the tile, two 64-float pieces per row
    typedef __attribute__((address_space(3))) unsigned char* lds_bptr;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(guide + (long long)b * 3 * HW), 0, (int)(3 * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (long long)b * Cmax * HW), 0, (int)((long long)Cmax * HW * 4), 0x00020000);
    const int xoffA = min(max(x0 - HALO + lane, 0), W - 1) * 4, xoffB = min(max(x0 - HALO + 64 + lane, 0), W - 1) * 4;   // replicate padding in x
    const unsigned tile_b = lds_addr(tile);
    constexpr int RPW = TR / 4;                                  // rows per wave
    // tiles that do not touch the left / right image border need no replication in x: 16-byte DMA, one instruction = two tile rows
    // (4x fewer instructions than the 4-byte form the border tiles need for per-element clamping)
    const bool interior = x0 - HALO >= 0 && x0 + 64 + HALO <= W;
    const int x4off = min(x0 - HALO + 4 * (lane & 31), W - 4) * 4;          // (columns >= 64 + 2 HALO of the 128-float row are never read)
    auto stage = [&](int p, int buf) {                           // plane p of the sequence [guide 0..2, mask 0..nch-1] -> buffer buf
        const bool is_g = p < 3;
        const int plane_off = (is_g ? p : p - 3) * (int)(HW * 4);
        unsigned dst = tile_b + (buf * TR + wave * RPW) * (TP * 4);
        asm volatile("" : "+s"(dst));
        if (interior) {
#pragma unroll 4
            for (int j = 0; j < RPW / 2; ++j) {
                const int gy = min(max(y0 - HALO + wave * RPW + 2 * j + (lane >> 5), 0), H - 1);   // replicate padding in y (per half-wave)
                const int voff = gy * W * 4 + x4off;
                if (is_g) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_bptr)(unsigned long long)(dst + j * (2 * TP * 4)), 16, voff, plane_off, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m, (lds_bptr)(unsigned long long)(dst + j * (2 * TP * 4)), 16, voff, plane_off, 0, 0);
            }
            return;
        }
#pragma unroll 4
        for (int rr = 0; rr < RPW; ++rr) {
            const int gy = min(max(y0 - HALO + wave * RPW + rr, 0), H - 1);                    // replicate padding in y (scalar)
            const int soff = plane_off + gy * W * 4;
            if (is_g) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_bptr)(unsigned long long)(dst + rr * (TP * 4)), 4, xoffA, soff, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_bptr)(unsigned long long)(dst + rr * (TP * 4) + 256), 4, xoffB, soff, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m, (lds_bptr)(unsigned long long)(dst + rr * (TP * 4)), 4, xoffA, soff, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m, (lds_bptr)(unsigned long long)(dst + rr * (TP * 4) + 256), 4, xoffB, soff, 0, 0);
            }
        }
    };

    f32x4 wall[ND][8];
#pragma unroll
    for (int di = 0; di < ND; ++di)
#pragma unroll
        for (int k = 0; k < 8; ++k) wall[di][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // taps of one staged plane: FN(di, k, float4 of the 4 neighbours); LDS reads as inline asm (a compiler-visible ds_read behind the
    // pending LDS-DMA of the NEXT plane would be preceded by s_waitcnt vmcnt(0))
    const int cb = HALO + 4 * tx;
    auto taps = [&](int buf, auto&& fn) {
        // opaque per plane: otherwise the ~50 per-lane LDS tap addresses (x 2 buffers) are hoisted out of the plane loop and compete
        // with the pinned weight registers (spills)
        int cbo = cb, tyo = ty;
        asm volatile("" : "+v"(cbo), "+v"(tyo));
        const unsigned base = tile_b + buf * (TR * TP * 4);
#pragma unroll
        for (int di = 0; di < ND; ++di) {
            const int d = dl.d[di];
            const int rrow[3] = {tyo + HALO - d, tyo + HALO, tyo + HALO + d};
            if ((d & 3) == 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {                    // one tap row at a time: at most 3 float4 in flight next to the 192 weight registers
                    const unsigned rowa = base + (rrow[r] * TP + cbo) * 4;
                    f32x4 L, M, R;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(L) : "v"(rowa - d * 4) : "memory");
                    if (r != 1) asm volatile("ds_read_b128 %0, %1" : "=v"(M) : "v"(rowa) : "memory");
                    asm volatile("ds_read_b128 %0, %1" : "=v"(R) : "v"(rowa + d * 4) : "memory");
                    if (r != 1) {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(L), "+v"(M), "+v"(R)::"memory");
                        fn(di, r == 0 ? 0 : 5, L);
                        fn(di, r == 0 ? 1 : 6, M);
                        fn(di, r == 0 ? 2 : 7, R);
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(L), "+v"(R)::"memory");
                        fn(di, 3, L);
                        fn(di, 4, R);
                    }
                    __builtin_amdgcn_sched_barrier(0);           // consume this row before the next row's reads are issued
                }
            } else {
                auto window = [&](auto shift_tag) {
                    constexpr int SH = decltype(shift_tag)::value;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const unsigned rowa = base + (rrow[r] * TP + cbo) * 4;
                        f32x4 L, M, R;
                        asm volatile("ds_read_b128 %0, %1" : "=v"(L) : "v"(rowa - 16) : "memory");
                        asm volatile("ds_read_b128 %0, %1" : "=v"(M) : "v"(rowa) : "memory");
                        asm volatile("ds_read_b128 %0, %1" : "=v"(R) : "v"(rowa + 16) : "memory");
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(L), "+v"(M), "+v"(R)::"memory");
                        const float win[12] = {L[0], L[1], L[2], L[3], M[0], M[1], M[2], M[3], R[0], R[1], R[2], R[3]};
#pragma unroll
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (r == 1 && dx == 0) continue;
                            const int k = (r == 0) ? dx + 1 : (r == 1 ? (dx < 0 ? 3 : 4) : dx + 6);
                            fn(di, k, f32x4{win[4 + SH * dx], win[5 + SH * dx], win[6 + SH * dx], win[7 + SH * dx]});
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if (d == 1) window(std::integral_constant<int, 1>{});
                else if (d == 2) window(std::integral_constant<int, 2>{});
                else window(std::integral_constant<int, 3>{});
            }
        }
    };

    const int np = 3 + nch;
    stage(0, 0);
    float* out_px = out + (long long)b * Cmax * HW + (long long)py * W + px;
    // phase 1 in its own loop (one body: with two different bodies in one rolled loop the 192 in-place accumulators were copied / spilled)
#pragma unroll 1
    for (int p = 0; p < 3; ++p) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of plane p have landed
        __builtin_amdgcn_s_barrier();                                // ... everyone's have, and everyone is done with plane p-1's buffer
        stage(p + 1, (p + 1) & 1);                                   // (np >= 4) streams in behind the taps below
        if (EXCEL_DBG(dbg) & 1) continue;
        // guide channel p:  z_t += -(I_nb - I)^2 k2_p   as fma(dv dv, -k2, z), dv = nb + (-ctr): bit-identical to the affinity kernel,
        // in forms that map onto packed VALU instructions without separate negations
        const f32x4 nk2 = -*reinterpret_cast<const f32x4*>(st_b + (long long)p * HW);
        f32x4 ctr;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ctr) : "v"(tile_b + (((p & 1) * TR + ty + HALO) * TP + cb) * 4) : "memory");
        const f32x4 nctr = -ctr;
        taps(p & 1, [&](int di, int k, const f32x4 nb) {
            const f32x4 dv = nb + nctr;
            wall[di][k] = __builtin_elementwise_fma(dv * dv, nk2, wall[di][k]);
        });
    }
    if (!(EXCEL_DBG(dbg) & 2)) {
        // aff_t = exp(z_t / 3 - m) / sum + pos_t : the affinity kernel's operations, one rounding each
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(st_b + 3 * HW), is4 = *reinterpret_cast<const f32x4*>(st_b + 4 * HW);
#pragma unroll
        for (int di = 0; di < ND; ++di)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                f32x4 e;
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = __expf(__fsub_rn(__fmul_rn(wall[di][k][j], 1.f / 3.f), m4[j]));
                const float ps = dl.pos_sm[di * 8 + k];
                wall[di][k] = __builtin_elementwise_fma(e, is4, f32x4{ps, ps, ps, ps});
            }
    }
    // phase 2: the weights are loop invariant
    for (int p = 3; p < np; ++p) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (p + 1 < np) stage(p + 1, (p + 1) & 1);
        if (EXCEL_DBG(dbg) & 4) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        taps(p & 1, [&](int di, int k, const f32x4 nb) { acc = __builtin_elementwise_fma(nb, wall[di][k], acc); });   // tap order, fused
        if (valid) *reinterpret_cast<f32x4*>(out_px + (long long)(p - 3) * HW) = acc;
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
static void par_aff_launch(const float* img, float* aff, const ParDil& dl, int B, int H, int W, float w1, hipStream_t st, bool compact) {
    hipLaunchKernelGGL(par_affinity_kernel<ND>, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(256), 0, st, img, aff, dl, H, W, w1, compact ? 1 : 0);
}
template <int ND>
static void par_it_launch(const float* aff, const float* in, float* out, const int* nchan, const ParDil& dl, int B, int Cmax,
                          int H, int W, hipStream_t st) {
    bool vec = (W % 4) == 0 && W >= 8 && ((((uintptr_t)aff | (uintptr_t)in | (uintptr_t)out) & 15) == 0);
    for (int i = 0; i < ND; ++i) vec = vec && ((dl.d[i] & 3) == 0 || dl.d[i] <= 3);
    int halo = 0;
    for (int i = 0; i < ND; ++i) halo = dl.d[i] > halo ? dl.d[i] : halo;
    halo = (halo + 3) / 4 * 4;
#ifdef EXCEL_DEV
    static const bool no_lds = getenv("EXCEL_PAR_NO_LDS") != nullptr;
#else
    const bool no_lds = false;
#endif
    if (vec && halo == 24 && !no_lds)
        hipLaunchKernelGGL((par_iterate_lds_kernel<ND, 24>), dim3(cdiv(W, 64), cdiv(H, 16), B), dim3(256), 0, st, aff, in, out, nchan, dl, Cmax, H, W);
    else if (vec && halo == 8 && !no_lds)
        hipLaunchKernelGGL((par_iterate_lds_kernel<ND, 8>), dim3(cdiv(W, 64), cdiv(H, 16), B), dim3(256), 0, st, aff, in, out, nchan, dl, Cmax, H, W);
    else if (vec)
        hipLaunchKernelGGL(par_iterate4_kernel<ND>, dim3(cdiv(W, 256), cdiv(H, 4), B), dim3(256), 0, st, aff, in, out, nchan, dl, Cmax, H, W);
    else
        hipLaunchKernelGGL(par_iterate_kernel<ND>, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(256), 0, st, aff, in, out, nchan, dl, Cmax, H, W);
}

// guide-recompute path: same preconditions as the LDS kernel
static bool par_guide_ok(const void* guide, const void* stats, const void* in, const void* out, int H, int W, const int* dil, int ndil, int* halo_out) {
    bool vec = (W % 4) == 0 && W >= 8 && ((((uintptr_t)guide | (uintptr_t)stats | (uintptr_t)in | (uintptr_t)out) & 15) == 0);
    int halo = 0;
    for (int i = 0; i < ndil; ++i) { vec = vec && ((dil[i] & 3) == 0 || dil[i] <= 3); halo = dil[i] > halo ? dil[i] : halo; }
    halo = (halo + 3) / 4 * 4;
    *halo_out = halo;
    // 8*ndil float4 weights stay in registers (6 dilations = 192 of 256 VGPRs); the scalar offsets of the LDS-DMA are 32-bit
    return vec && (halo == 24 || halo == 8) && ndil <= 6 && (long long)H * W * 4 * 64 < (1LL << 31);
}
template <int ND>
static void par_guide_launch(const float* guide, const float* stats, const float* in, float* out, const int* nchan, const ParDil& dl, int B,
                             int Cmax, int H, int W, int halo, hipStream_t st) {
    const dim3 grid(cdiv(W, 64), cdiv(H, 16), B);
    int dbg = 0;
#ifdef EXCEL_DEV
    static const int env_dbg = getenv("EXCEL_PAR_DBG") ? atoi(getenv("EXCEL_PAR_DBG")) : 0;
    dbg = env_dbg;
#endif
    if (halo == 24) hipLaunchKernelGGL((par_iterate_guide_kernel<ND, 24>), grid, dim3(256), 0, st, guide, stats, in, out, nchan, dl, Cmax, H, W, dbg);
    else hipLaunchKernelGGL((par_iterate_guide_kernel<ND, 8>), grid, dim3(256), 0, st, guide, stats, in, out, nchan, dl, Cmax, H, W, dbg);
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
                              hipStream_t st, int compact) {
    ProfScope prof__(PROF_PAR_AFFINITY, st);
    ParDil dl;
    int rc = make_dil(dil, ndil, w1, w2, &dl);
    if (rc) return rc;
#define CALL(N) par_aff_launch<N>(img, aff, dl, B, H, W, w1, st, compact != 0)
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

// One Jacobi step with the affinities recomputed from (guide, stats); returns 1 when the shape is not supported (caller streams
// the 48 affinity planes instead).  w1 / w2 enter through the position term only (the colour term's w1 is inside stats).
int excel_par_guide_supported(const float* guide, const float* stats, const float* in, const float* out, int H, int W, const int* dil, int ndil) {
    int halo;
    return par_guide_ok(guide, stats, in, out, H, W, dil, ndil, &halo) ? 1 : 0;
}
int excel_launch_par_iterate_guide(const float* guide, const float* stats, const float* in, float* out, const int* nchan, int B, int Cmax,
                                   int H, int W, const int* dil, int ndil, float w1, float w2, hipStream_t st) {
    int halo;
    if (!par_guide_ok(guide, stats, in, out, H, W, dil, ndil, &halo)) return 1;
    ProfScope prof__(PROF_PAR_ITERATE, st);
    ParDil dl;
    int rc = make_dil(dil, ndil, w1, w2, &dl);
    if (rc) return rc;
#define CALL(N) par_guide_launch<N>(guide, stats, in, out, nchan, dl, B, Cmax, H, W, halo, st)
    ND_SWITCH(ndil, CALL)
#undef CALL
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
    // scalar elements in front of the first 16-byte boundary (same for both pointers), or -1: no common alignment -> scalar path
    const int mg = (int)((16 - ((uintptr_t)gt & 15)) & 15), mp = (int)((16 - ((uintptr_t)pred & 15)) & 15);
    const int head = (mg == mp) ? mg : -1;
    const int blocks = (int)((cdivl(n, 256 * 16) < 2048) ? cdivl(n, 256 * 16) : 2048);
    hipLaunchKernelGGL(confusion_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, gt, pred, n, nc, hist, head);
    EXCEL_CHECK_LAUNCH("confusion");
    return EXCEL_OK;
}
