// fp32 GEMM on the CDNA4 f32-input matrix core (v_mfma_f32_32x32x2_f32): exact-f32 numerics
// (bitwise an fmaf chain) at the f32 matrix rate.  The reference computes everything in fp32
// (clip/build_model.py:72 disables the fp16 conversion) and the CAM gate is 1e-3 after a
// min-max normalisation, so the ViT's GEMMs stay f32-in/f32-accumulate.
//
//   C[M,N] = act(A[M,K] * op(B) + bias) + residual         (batched over blockIdx.z)
//   B_KMAJOR: B is [N,K] row-major (torch nn.Linear weight)  -> "NT"
//   else    : B is [K,N] row-major                           -> "NN"
//
// Tiling: 128x128 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32,
// BK = 32 staged through LDS (register-staged double buffer, one barrier per K step).
// Fragment trick: lane (r = lane&31, kh = lane>>5) reads ONE ds_read_b128 = 4 consecutive k
// (k = kk*8 + kh*4 + j) and feeds 4 MFMAs; the k-pairing of an MFMA step is {kk*8+j, kk*8+4+j},
// legal because A and B use the same pairing and fp32 addition order inside a step is fixed.
#include "common.h"
#include "excel_internal.h"

namespace EXCEL_SPLIT_NS {     // compiled once per 16-bit split type (excel_internal.h, build.py)

#define BM 128
#define BN 128
#define BK 32
#define APITCH 36   // floats; 144 B rows -> 16-B slot index 9*row mod 16: conflict-free b128 reads
#define BPITCH_N 128

template <bool B_KMAJOR>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float As[2][BM * APITCH];
    __shared__ __attribute__((aligned(16))) float Bs[2][B_KMAJOR ? BN * APITCH : BK * BPITCH_N];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, kh = lane >> 5;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = id / tiles_n, tn = id % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;

    const int z1 = z / p.zdiv, z2 = z % p.zdiv;
    const float* __restrict__ A = p.A + (long long)z1 * p.sA + (long long)z2 * p.sA2;
    const float* __restrict__ B = p.B + (long long)z1 * p.sB + (long long)z2 * p.sB2;

    // global -> register staging map: 128 rows x 8 float4 per operand tile (K-major), 4 per thread
    const int lrow = tid >> 3;        // 0..31 (+32*i)
    const int lc4 = tid & 7;          // float4 column
    // NN B tile: 32 k-rows x 32 float4
    const int brow = tid >> 5;        // 0..7 (+8*i)
    const int bc4 = tid & 31;

    f32x4 ra[4], rb[4];

    auto load_tile = [&](int kt) {
        const int k = kt * BK + lc4 * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = min(m0 + lrow + 32 * i, p.M - 1);
            if (k < p.Kld)
                ra[i] = *reinterpret_cast<const f32x4*>(A + (long long)row * p.lda + k);
            else
                ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (B_KMAJOR) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = min(n0 + lrow + 32 * i, p.N - 1);
                if (k < p.Kld)
                    rb[i] = *reinterpret_cast<const f32x4*>(B + (long long)row * p.ldb + k);
                else
                    rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kr = kt * BK + brow + 8 * i;
                const int n = n0 + bc4 * 4;
                if (kr < p.K && n < p.N)
                    rb[i] = *reinterpret_cast<const f32x4*>(B + (long long)kr * p.ldb + n);
                else
                    rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(&As[buf][(lrow + 32 * i) * APITCH + lc4 * 4]) = ra[i];
        if (B_KMAJOR) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + 32 * i) * APITCH + lc4 * 4]) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4*>(&Bs[buf][(brow + 8 * i) * BPITCH_N + bc4 * 4]) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (p.Kld + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* as = As[cur];
        const float* bs = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(&as[(wm * 64 + i * 32 + r) * APITCH + kk * 8 + kh * 4]);
            if (B_KMAJOR) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    b[j] = *reinterpret_cast<const f32x4*>(&bs[(wn * 64 + j * 32 + r) * APITCH + kk * 8 + kh * 4]);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        b[j][e] = bs[(kk * 8 + kh * 4 + e) * BPITCH_N + wn * 64 + j * 32 + r];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: bias -> activation -> (+ residual) -> store
    const float* bias = p.bias ? p.bias + (long long)z1 * p.sBias : nullptr;
    const float* res = p.res ? p.res + (long long)z1 * p.sR : nullptr;
    float* C = p.C + (long long)z1 * p.sC + (long long)z2 * p.sC2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + r;
        if (col >= p.N) continue;
        const float bv = bias ? bias[col] : 0.f;
        // head-major split of the packed q|k|v projection: col -> (type, head, d)
        int qt = 0, qh = 0, qd = 0;
        if (p.out_mode == GEMM_OUT_QKV_HEADMAJOR) {
            const int D = p.heads * p.hd;
            qt = col / D;
            qh = (col % D) / p.hd;
            qd = col % p.hd;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + c32_row(e, lane);
                if (row >= p.M) continue;
                float v = acc[i][j][e] * p.alpha + bv;
                if (p.act == GEMM_ACT_QUICKGELU) v = v * (1.f / (1.f + __expf(-1.702f * v)));
                else if (p.act == GEMM_ACT_RELU) v = fmaxf(v, 0.f);
                if (res) v += res[(long long)row * p.ldr + col];
                if (p.out_mode == GEMM_OUT_SPLIT_BF16) {
                    // split-bf16 output [rows][2][ldc]: plane width ldc, z2 offsets columns (in bf16 elements)
                    // (z2 offsets columns by sC2 elements of the logical row; blocked hi/lo layout, see split_off)
                    split_t* o = reinterpret_cast<split_t*>(p.C + (long long)z1 * p.sC) + (long long)row * 2 * p.ldc +
                                split_off((int)(z2 * p.sC2) + col, 0);
                    const split_t hi = split_hi(v);
                    o[0] = hi;
                    o[32] = split_hi(v - (float)hi);
                } else if (p.out_mode == GEMM_OUT_QKV_HEADMAJOR) {
                    const int b = row / p.tokN, n = row % p.tokN;
                    C[((((long long)b * 3 + qt) * p.heads + qh) * p.tokN + n) * p.hd + qd] = v;
                } else {
                    C[(long long)row * p.ldc + col] = v;
                }
            }
        }
    }
}

int excel_launch_gemm(const GemmArgs& p, bool b_kmajor, int batch, hipStream_t stream) {
    ProfScope prof__(g_excel_prof_gemm_cat >= 0 ? g_excel_prof_gemm_cat : (b_kmajor ? PROF_GEMM_NT : PROF_GEMM_NN), stream, 2.0 * p.M * (double)p.N * p.K * batch);
    EXCEL_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && batch > 0, "gemm: bad shape M=%d N=%d K=%d batch=%d", p.M, p.N, p.K, batch);
    EXCEL_CHECK_ARG((p.lda % 4) == 0 && (p.Kld % 4) == 0 && p.Kld <= p.lda, "gemm: lda/Kld must be multiples of 4 (lda=%d Kld=%d)", p.lda, p.Kld);
    EXCEL_CHECK_ARG((((uintptr_t)p.A | (uintptr_t)p.B) & 15) == 0, "gemm: A/B must be 16-byte aligned");
    EXCEL_CHECK_ARG((p.sA % 4) == 0 && (p.sB % 4) == 0 && (p.sA2 % 4) == 0 && (p.sB2 % 4) == 0 && p.zdiv >= 1,
                    "gemm: batch strides must be multiples of 4, zdiv >= 1");
    if (b_kmajor) {
        EXCEL_CHECK_ARG((p.ldb % 4) == 0 && p.Kld <= p.ldb, "gemm(NT): ldb must be a multiple of 4 and >= Kld");
    } else {
        EXCEL_CHECK_ARG((p.ldb % 4) == 0 && (p.N % 4) == 0, "gemm(NN): ldb and N must be multiples of 4");
    }
    const int tiles = cdiv(p.M, BM) * cdiv(p.N, BN);
    dim3 grid(tiles, 1, batch);
    if (b_kmajor)
        hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, stream, p);
    EXCEL_CHECK_LAUNCH("gemm_f32");
    return EXCEL_OK;
}

}  // namespace EXCEL_SPLIT_NS
