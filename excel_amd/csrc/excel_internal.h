// Internal (C++) launch interfaces shared between the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>

enum { GEMM_ACT_NONE = 0, GEMM_ACT_QUICKGELU = 1, GEMM_ACT_RELU = 2 };
enum { GEMM_OUT_PLAIN = 0, GEMM_OUT_QKV_HEADMAJOR = 1, GEMM_OUT_SPLIT_BF16 = 2 };

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   // [N] or null
    const float* res;    // [M,ldr] or null (may alias C: x += ...)
    int M, N, K;         // K = true reduction length (rows of B guarded by it in NN mode)
    int Kld;             // K rounded up to a multiple of 4 that A (and B in NT mode) can be read to;
                         // the operands must hold zeros (or finite*0-safe data with the other side zero) there
    int lda, ldb, ldc, ldr;
    long long sA, sB, sC, sR, sBias;   // batch strides in elements, applied to z / zdiv
    int zdiv;                          // blockIdx.z = z1*zdiv + z2 (zdiv >= 1)
    long long sA2, sB2, sC2;           // strides applied to z % zdiv
    int act;
    int out_mode;
    int tokN, heads, hd;  // GEMM_OUT_QKV_HEADMAJOR: C is [B,3,heads,tokN,hd]
    float alpha;
};


// bf16x3 GEMM (gemm_bf16x3.hip): A, B are "split" tensors [rows][2][K] bf16 (hi plane, lo plane)
struct GemmBfArgs {
    const unsigned short* A;   // [M][2][K]  (lda = elements per row >= 2K)
    const unsigned short* B;   // [N][2][K]
    float* C;                  // fp32 output (plain / head-major)
    unsigned short* Cs;        // split output [M][2][N] (GEMM_OUT_SPLIT_BF16)
    const float* bias;
    const float* res;
    int M, N, K;
    int lda, ldb, ldc, ldr;
    int act, out_mode;
    int tokN, heads, hd;
    unsigned short* qkv_split;   // GEMM_OUT_QKV_HEADMAJOR: also write q|k|v head-major in split format [B,3,H,N][2][hd] (may be null)
    int dbg;                     // dev only: bit0 = skip MFMA/LDS-read work, bit1 = skip the staging loads after the first tile
    int batch;                   // >= 1: blockIdx.y; operands advance by sA / sB bf16 elements, C by sC floats, Cs by sCs bf16 elements
    long long sA, sB, sC, sCs;
    int mix_tall, mix_short;     // set by the launcher (mixed-height 320x256 / 256x256 row tiles, gemm_bf16x3.hip); 0 = uniform tiles
    int mix_first;               // short tiles dispatched FIRST in every XCD's chunk (the rest follow the tall ones): staggers the epilogues
    // "f16x2" (IEEE-half split type only): the caller guarantees that B's lo plane is all zero (fp16-valued weights) - the a.hi x b.lo
    // products are skipped (bit-identical results, 2 MFMAs per product).  Bh (optional): the same weights as a plain half matrix
    // [N][ldbh >= K], the four-wave kernel's compact operand (gemm_w4x2.hip)
    int w_lo_zero;
    const unsigned short* Bh;
    int ldbh;
};


int excel_launch_trans_mat_sym(const float* W, float* T, float* Tsym, float* cs, int B, int P, hipStream_t st);
int excel_launch_cls_compact(const float* onehot, int B, int F, int Smax, int* cls_idx, int* ncls, int* nchan, hipStream_t st);
int excel_launch_bbox_mask(const float* attr, const int* cls_idx, const int* ncls, int B, int g, int F, int Smax, double thre,
                           float* v_out, unsigned char* mask_out, hipStream_t st);
int excel_launch_matvec(const float* T, const float* v, const int* ncls, float* u, int B, int P, int Smax, hipStream_t st);
int excel_launch_cam_upsample_bkg(const float* r, const int* ncls, float* rn, float* cams, int B, int g, int Smax, int H, int W,
                                  int zero_unused, hipStream_t st);
struct TileGeo;
int excel_launch_par_affinity(const float* img, float* aff, const TileGeo& geo, int total_tiles, const int* dil, int ndil, float w1, float w2,
                              hipStream_t st, int compact = 0);
int excel_par_guide_supported(const void* guide, const void* stats, const void* in, const void* out, int Cmax, long long max_plane, int Wp,
                              const int* dil, int ndil);
int excel_launch_par_iterate_guide(const float* guide, const float* stats, const float* in, float* out, const int* nchan, int Cmax,
                                   const TileGeo& geo, int total_tiles, const int* dil, int ndil, float w1, float w2, hipStream_t st);
int excel_launch_par_iterate(const float* aff, const float* in, float* out, const int* nchan, int B, int Cmax, int H, int W,
                             const int* dil, int ndil, hipStream_t st);
int excel_launch_bilinear_ac_ragged(const float* in, float* out, int h, int w, const TileGeo& geo, int total_tiles, hipStream_t st);
int excel_launch_argmax_label_ragged(const float* cams, const int* nchan, const int* cls_idx, int Smax, int Cmax, const TileGeo& geo,
                                     int total_tiles, unsigned char* lab8, hipStream_t st);
int excel_launch_cam_upsample_bkg_ragged(const float* r, const int* ncls, float* rn, float* cams, int g, int Smax, const TileGeo& geo,
                                         int total_tiles, int zero_unused, hipStream_t st);
int excel_launch_normalize_resize_u8_ragged(const unsigned char* hwc, float* out, const TileGeo& geo, int S, const double* mean, const double* stdv,
                                            hipStream_t st);
int excel_launch_bilinear_ac(const float* in, float* out, int planes, int h, int w, int H, int W, hipStream_t st);
int excel_launch_argmax_label(const float* cams, const int* nchan, const int* cls_idx, int B, int Smax, int Cmax, long long HW,
                              unsigned char* lab8, long long* lab64, hipStream_t st);
int excel_launch_confusion(const unsigned char* gt, const unsigned char* pred, long long n, int nc, unsigned long long* hist,
                           hipStream_t st);
int excel_launch_attr_aggregate(const float* text, const float* bank, int F, int T, int C, int K, int drop, float* out,
                                hipStream_t st);
int excel_launch_bilinear_resize(const float* in, float* out, long long planes, int h, int w, int H, int W, int align_corners,
                                 hipStream_t st);
int excel_launch_flip_max_normalize(const float* attr, float* out, int B, int g, int F, hipStream_t st);
int excel_launch_lam_scale_accumulate(const float* maps, float* acc, int B, int g, int F, int H, int W, int init, hipStream_t st);
int excel_launch_plane_minmax_normalize(float* lam, long long planes, long long HW, hipStream_t st);
int excel_launch_seg_scale_accumulate(const float* segs, float* acc, int B, int nc, int h, int w, int H, int W, int flip_mean, int init,
                                      float scale, hipStream_t st);
int excel_launch_denormalize(const float* img, unsigned char* out8, float* outf, int B, long long HW, const float* mean, const float* stdv,
                             hipStream_t st);
int excel_launch_normalize_u8(const unsigned char* hwc, float* out, int B, long long HW, const double* mean, const double* stdv, hipStream_t st);
int excel_launch_lam_to_label(const float* cam, const float* cls, const int* box, int B, int F, int H, int W, float bkg, float high, float low,
                              int ignore_mid, int ignore, float* valid, unsigned char* lab, hipStream_t st);
// training step (train.hip)
size_t excel_train_losses_ws_bytes(int B, int nc, int H, int W);
int excel_launch_train_losses(const float* seg, const float* attn_pred, const unsigned char* pseudo, const unsigned char* aff_labels, int B, int nc,
                              int g_h, int g_w, int H, int W, int radius, int ignore, float w_seg, float w_diver, float* losses, float* d_seg,
                              float* d_attn_pred, void* ws, hipStream_t st);
// LVC side (lvc.hip)
size_t excel_feature_affinity_ws_bytes(int B, int C, int P);
int excel_launch_feature_affinity(const float* feats, int B, int C, int P, float beta, float gamma, int mode, float* out, void* ws,
                                  hipStream_t st);
size_t excel_attn_select_ws_bytes(int B, int n_layers);
int excel_launch_attn_select_mean(const float* attn, int Lw, int B, int N, int first_layer, int n_layers, const float* seg_attn,
                                  float* out, void* ws, hipStream_t st);

// ---------------------------------------------------------------- the split-type dependent launchers, once per 16-bit type
namespace excel_bf16 {
#include "excel_split_api.inc"
}
namespace excel_f16 {
#include "excel_split_api.inc"
}
#ifdef EXCEL_SPLIT_F16
#define EXCEL_SPLIT_NS excel_f16
#else
#define EXCEL_SPLIT_NS excel_bf16
#endif
// unqualified calls (decoder.hip, lvc.hip, train.hip, ...: exact-fp32 paths that never write split planes) mean the bf16 build
using namespace excel_bf16;
