/* libexcel_hip -- C ABI of the MI355X-native ExCEL training-free CAM + affinity + PAR hot path.
 *
 * The reference (zwyang6/ExCEL) is pure Python on PyTorch: it has no FFI/plugin boundary, its drop-in
 * boundary is the Python call surface (model_excel / clip / affutils / PAR / evaluate).  excel_amd/ keeps
 * that surface and binds THIS library underneath it with ctypes.  Every entry point
 * below cites the reference interface it replaces (paths relative to the reference repo).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless marked (host); tensors are dense row-major fp32 unless noted
 *   - every call takes an explicit hipStream_t (passed as void*), launches asynchronously, never syncs,
 *     never allocates (callers own all buffers; *_workspace_bytes tell how much scratch to bring),
 *     except excel_vit_create / excel_vit_destroy
 *   - return 0 on success, <0 on error (-1 bad argument, -2 launch failure, -3 allocation failure);
 *     excel_last_error() returns a thread-local message
 *   - B images, S x S input, g = S/patch, P = g*g patches, N = P+1 tokens, D = width, H = heads (D/H must be 64),
 *     C = out_dim, T text rows, F foreground classes, L layers
 */
#ifndef EXCEL_HIP_H
#define EXCEL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* excel_last_error(void);
int excel_abi_version(void);
/* (host) id of the sources this library was compiled from: first 16 hex digits of the sha256 over csrc/*.hip, csrc/*.h and this header
 * (excel_amd/build.py:source_id; "-dev" appended for an EXCEL_DEV build, "unstamped" when compiled outside build.py).  The harness
 * compares it with the sources next to it so that a stale library is never measured. */
const char* excel_build_id(void);

/* ------------------------------------------------------------------ building blocks (exported for tests / reuse) */

/* C = act(A . op(B) + bias) + residual on the f32-input matrix core (exact-fp32 numerics).
 * b_kmajor=1: B is [N,K] (torch nn.Linear weight layout); 0: B is [K,N].  act: 0 none, 1 QuickGELU
 * (clip/clip_surgery_model.py:280-282).  Batched: `batch` problems with element strides sA/sB/sC.
 * Replaces the cuBLAS GEMMs behind nn.Linear / torch.matmul on the path. */
int excel_gemm_f32(const float* A, const float* Bm, float* C, const float* bias, const float* residual,
                   int M, int N, int K, int lda, int ldb, int ldc, int ldr, int b_kmajor, int act,
                   int batch, long long sA, long long sB, long long sC, long long sR, void* stream);

/* "bf16x3" building blocks: excel_split_bf16 turns fp32 [rows,K] into the split operand format [rows][2][K] bf16
 * (hi plane, lo plane; same number of bytes); excel_gemm_bf16x3 computes C = act(A.W^T + bias) + residual from two split
 * operands with three bf16 MFMAs per product (fp32 accumulate).  C is fp32 [M,N], or a split tensor when split_out. */
int excel_split_bf16(const float* in, void* out, long long rows, int K, void* stream);
int excel_gemm_bf16x3(const void* A_split, const void* W_split, float* C, const float* bias, const float* residual,
                      int M, int N, int K, int act, int split_out, void* stream);
/* The same two building blocks with IEEE-half planes ("f16x3": hi = half(x), lo = half(x - hi); 11 + 11 mantissa bits where lo stays
 * a normal half, range 65 504; v_mfma_f32_16x16x32_f16 in the GEMM, same layouts and rate). */
int excel_split_f16(const float* in, void* out, long long rows, int K, void* stream);
int excel_gemm_f16x3(const void* A_split, const void* W_split, float* C, const float* bias, const float* residual,
                     int M, int N, int K, int act, int split_out, void* stream);
/* "f16x2": the f16x3 GEMM for fp16-VALUED weights - two matrix-core instructions per product.  The published CLIP archives store
 * half-precision parameters and the reference loads them into its fp32 model unchanged (clip/build_model.py:72, clip/clip.py:138-154),
 * so the lo plane of every nn.Linear weight is exactly zero and the a.hi x w.lo pass of the three-product scheme multiplies zeros; it
 * is not issued.  The results are bit-identical to excel_gemm_f16x3 on the same operands.
 *   excel_pack_f16: fp32 [rows,K] -> the hi plane as a plain half matrix [rows,K]; adds to *inexact_dev (device, zero it first) the
 *                   number of elements that are NOT representable in IEEE half - the mode's precondition is that count == 0.
 *   excel_gemm_f16x2: W_split as excel_split_f16 wrote it (its lo plane must be all zero: not checked here); W_half (optional, may be
 *                   NULL) = the excel_pack_f16 matrix of the same weights: the large-tile kernel then streams half the weight bytes. */
int excel_pack_f16(const float* in, void* out, long long rows, int K, unsigned long long* inexact_dev, void* stream);
int excel_gemm_f16x2(const void* A_split, const void* W_split, const void* W_half, float* C, const float* bias, const float* residual,
                     int M, int N, int K, int act, int split_out, void* stream);

/* LayerNorm over the last dim, fp32, eps as given (clip/clip_surgery_model.py:271-277). */
int excel_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int D, float eps, void* stream);

/* ------------------------------------------------------------------ ViT "surgery" forward */

typedef struct {
    int width, layers, heads, patch, out_dim;
    int n_surgery;   /* blocks [layers-n_surgery, layers) use q-q/k-k/v-v attention; reload_self_attn(layers=6) -> 5
                        (clip/clip_surgery_model.py:396-405) */
    int pos_grid;    /* side of the stored positional grid (positional_embedding has 1 + pos_grid^2 rows) */
} excel_vit_config;

typedef struct {
    const float *ln1_w, *ln1_b;
    const float *in_proj_w, *in_proj_b;    /* [3D,D], [3D]  rows q|k|v (nn.MultiheadAttention.in_proj_* == Attention.qkv.*) */
    const float *out_proj_w, *out_proj_b;  /* [D,D], [D] */
    const float *ln2_w, *ln2_b;
    const float *fc1_w, *fc1_b;            /* mlp.c_fc   [4D,D], [4D] */
    const float *fc2_w, *fc2_b;            /* mlp.c_proj [D,4D], [D] */
} excel_vit_block_weights;

typedef struct {
    const float* conv1_w;    /* [D,3,patch,patch] */
    const float* class_emb;  /* [D] */
    const float* pos_emb;    /* [1+pos_grid^2, D] */
    const float *ln_pre_w, *ln_pre_b, *ln_post_w, *ln_post_b;
    const float* proj;       /* [D, out_dim] */
    const excel_vit_block_weights* blocks;   /* (host) array of `layers` entries of device pointers */
} excel_vit_weights;

typedef struct excel_vit* excel_vit_t;

/* Binds device weight pointers (NOT copied, must outlive the handle) and prepares derived weights
 * (proj^T; bilinearly resized positional grids are cached per g on first use, clip_surgery_model.py:426-435).
 * Replaces ExCEL_CLIP.visual construction + reload_self_attn (clip/clip_surgery_model.py:396-416). */
int excel_vit_create(const excel_vit_config* cfg, const excel_vit_weights* w, excel_vit_t* out);
void excel_vit_destroy(excel_vit_t h);

/* GEMM numerics of the linear layers (and attention products) of this handle:
 *   0 = exact fp32 (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak);
 *   1 = "bf16x3": operands as bf16 hi+lo planes, 3 bf16 MFMAs per product, 16 mantissa bits per operand at the fp32 exponent range -
 *       the fastest mode; its CAM error is ~12x that of fp32 arithmetic (1e-5 on well-conditioned weights);
 *   2 = "f16x3": the same scheme on IEEE-half planes (22 mantissa bits where lo stays normal): fp32-grade results (within 1.4x of fp32
 *       arithmetic's own error on every network measured), ~2.5 % slower than mode 1 (the part is power-limited), values beyond 65 504
 *       overflow LOUDLY (hi = inf, lo = -inf: every product they enter is a NaN, never a finite wrong number);
 *   3 = "f16x2": mode 2 with the nn.Linear GEMMs on two MFMAs per product (excel_gemm_f16x2 above) - available when every weight
 *       matrix of the handle is fp16-valued (excel_vit_weights_fp16_exact; true for every published CLIP archive), refused with
 *       EXCEL_ERR_ARG otherwise.  Bit-identical to mode 2 on such weights, and the fastest mode: a third of the GEMMs' matrix-core
 *       work is gone.  The attention products (activation x activation) stay three-product.
 * Default 0, or the mode named by the environment variable EXCEL_GEMM_MODE (bf16x3 | f16x3 | f16x2) at create time.  Switching between
 * 1 and 2/3 re-splits the weights (synchronises the device). */
int excel_vit_set_gemm_mode(excel_vit_t h, int mode);
int excel_vit_get_gemm_mode(excel_vit_t h);
/* 1 when every GEMM weight of the handle (in_proj, out_proj, fc1, fc2 of every block, conv1, proj) is exactly representable in IEEE
 * half, 0 when not, < 0 on error.  The first call packs the weights (one pass over them, synchronises); the answer is cached. */
int excel_vit_weights_fp16_exact(excel_vit_t h);

size_t excel_vit_workspace_bytes(excel_vit_t h, int B, int S);

/* VisionTransformer.forward + generate_clip_fts (clip/clip_surgery_model.py:419-448, clip/clip.py:348-358).
 *   img            [B,3,S,S]
 *   image_features [B,N,C]  token-axis L2-normalised (clip.py:353)                         (optional when x_raw is given)
 *   x_raw          [B,N,C]  ln_post(x) @ proj before the normalisation                      (optional, may be NULL)
 *   w_aff          [B,P,P]  mean over the last `aff_layers` layers of attn[:,1:,1:] -- exactly what
 *                           refine_cams_with_aff consumes (utils/affutils.py:180,197); block weights are head-MEAN
 *                           for nn.MultiheadAttention blocks, head-SUM for surgery blocks    (optional)
 *   attn_out       [n_attn_out,B,N,N] per-layer weights of the LAST n_attn_out layers (0..L) (optional)
 *   feats_out      [L,B,N,D] per-block original-path features ("all_feats", clean copies)   (optional)
 */
int excel_vit_forward(excel_vit_t h, const float* img, int B, int S, void* workspace, size_t workspace_bytes,
                      float* image_features, float* x_raw, float* w_aff, int aff_layers,
                      float* attn_out, int n_attn_out, float* feats_out, void* stream);

/* Same, with the LVC cue of Attention.forward's `ex_feats` branch (clip/clip_surgery_model.py:127-141):
 *   ex_attn [B,P,P] (= excel_feature_affinity(ex_feats, mode 1)) is added to attn[:, :, 1:, 1:] of EVERY head of every
 *   surgery block before the head sum; NULL = the ex_feats=None branch (identical to excel_vit_forward).
 *   flags: EXCEL_VIT_FEATS_AS_REFERENCE = feats_out holds what the reference's decoder receives: the reference stacks
 *          its per-block list after the forward, and in-place updates (clip_surgery_model.py:317,319,329,442) have by then
 *          rewritten the entries of the last single-path block (= final new-path x incl. the cls swap) and of every
 *          surgery block but the last (= x_ori + the next block's attention residual).  Default: clean block outputs. */
#define EXCEL_VIT_FEATS_AS_REFERENCE 1
int excel_vit_forward_ex(excel_vit_t h, const float* img, int B, int S, void* workspace, size_t workspace_bytes,
                         float* image_features, float* x_raw, float* w_aff, int aff_layers,
                         float* attn_out, int n_attn_out, float* feats_out, const float* ex_attn, int flags, void* stream);

/* Token affinity of decoder features, shared by attn_pred (model/model_excel.py:70-76) and the ex_feats branch
 * (clip/clip_surgery_model.py:128-137):  feats [B,C,P] -> F.normalize over C -> sim = f^T f [B,P,P]
 *   -> z = (sim - mean(sim over the WHOLE batch tensor) * beta) * gamma
 *   mode 0: out = sigmoid(z)                                  (attn_pred: beta 1, gamma 3)
 *   mode 1: z < 0 -> -inf, out = softmax(z, dim=-1)           (ex_attn:   beta 1, gamma 3)                       */
size_t excel_feature_affinity_workspace_bytes(int B, int C, int P);
int excel_feature_affinity(const float* feats, int B, int C, int P, float beta, float gamma, int mode, float* out,
                           void* workspace, void* stream);

/* ------------------------------------------------------------------ decoder head
 * SegFormerHead fuse (model/segformer_head.py:47-77: per ViT layer Linear(D,E) -> ReLU -> Linear(E,E) on the patch
 * tokens, channel concat, 1x1 conv L*E -> E) and DecoderTransformer (model/decoder/TransDecoder.py:105-124: dec_layers
 * pre-LN blocks of MHA(heads) + QuickGELU MLP(4E), then the 1x1 linear_pred E -> num_classes), exact fp32.
 * Weight pointers are device pointers in the reference modules' state_dict layout (Linear: [out,in]; 1x1 conv:
 * [out,in,1,1] == [out,in]) and must stay valid for the life of the handle.                                    */
typedef struct excel_decoder* excel_decoder_t;
typedef struct { int vit_layers, vit_width, embed, dec_layers, heads, num_classes; } excel_decoder_config;
typedef struct { const float *proj_w, *proj_b, *proj2_w, *proj2_b; } excel_fuse_layer_weights;
typedef struct {
    const float *ln1_w, *ln1_b, *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} excel_decoder_block_weights;
typedef struct {
    const excel_fuse_layer_weights* fuse;          /* [vit_layers] */
    const float *fuse_w, *fuse_b;                  /* linear_fuse [E, L*E], [E] */
    const excel_decoder_block_weights* blocks;     /* [dec_layers] */
    const float *pred_w, *pred_b;                  /* linear_pred [num_classes, E], [num_classes] */
} excel_decoder_weights;
int excel_decoder_create(const excel_decoder_config* cfg, const excel_decoder_weights* w, excel_decoder_t* out);
void excel_decoder_destroy(excel_decoder_t h);
size_t excel_decoder_workspace_bytes(excel_decoder_t h, int B, int g);
/* all_feats [L,B,N,D] (excel_vit_forward's feats_out; N = g*g+1, the cls row is skipped as model/model_excel.py:60 does)
 *   -> attn_fts_out [B,E,g,g] (optional; model_excel.py:64-65) and seg_out [B,num_classes,g,g] (optional; :68).           */
int excel_decoder_forward(excel_decoder_t h, const float* all_feats, int B, int g, void* workspace, size_t workspace_bytes,
                          float* attn_fts_out, float* seg_out, void* stream);

/* ------------------------------------------------------------------ CLIP text tower (one-time text bank)
 * encode_text (clip/clip_surgery_model.py:551-564): tokens [B,context_length] int32 -> token + positional embedding ->
 * `layers` causal pre-LN blocks (nn.MultiheadAttention + QuickGELU MLP) -> ln_final -> row of the EOT token (first arg-max of
 * the ids) @ text_projection [width, embed_dim] -> out [B, embed_dim].  Exact fp32.  Weights: device pointers, state_dict layout. */
typedef struct excel_text* excel_text_t;
typedef struct { int vocab_size, context_length, width, layers, heads, embed_dim; } excel_text_config;
typedef struct {
    const float *token_embedding, *positional_embedding, *ln_final_w, *ln_final_b, *text_projection;
    const excel_decoder_block_weights* blocks;     /* [layers] */
} excel_text_weights;
int excel_text_create(const excel_text_config* cfg, const excel_text_weights* w, excel_text_t* out);
void excel_text_destroy(excel_text_t h);
size_t excel_text_workspace_bytes(excel_text_t h, int B);
int excel_text_encode(excel_text_t h, const int32_t* tokens, int B, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* encode_text_with_prompt_ensemble's reduction (clip/clip.py:262-266): emb [n,E] -> rows normalised, mean, normalised -> out [E]. */
int excel_prompt_ensemble(const float* emb, int n, int E, float* out, void* stream);

/* ------------------------------------------------------------------ training iteration of the decoder
 * Losses of scripts/train_voc.py:202-215 and their gradients:
 *   seg [B,nc,g_h,g_w] -> bilinear (align_corners=False) to (H,W) -> get_seg_loss (model/losses.py:4-18) against pseudo [B,H,W] u8
 *   attn_pred [B,P,P] -> get_aff_loss (model/losses.py:20-31) against cams_to_affinity_label(pseudo, get_mask_by_radius(g,g,radius))
 *                        (utils/camutils.py:438-476; nearest down-sampling by H/g_h, 255 outside the window / on ignored tokens)
 *   losses[0] = seg_loss, losses[1] = aff ("diver") loss (device floats); d_seg, d_attn_pred = gradients of
 *   w_seg*seg_loss + w_diver*aff_loss.  aff_labels (NULL = pseudo): the map the affinity labels are built from -- the reference
 *   switches it to the arg-max of the up-sampled seg logits after 24 000 iterations (train_voc.py:210).  Fixed-order reductions. */
size_t excel_train_losses_workspace_bytes(int B, int nc, int H, int W);
int excel_train_losses(const float* seg, const float* attn_pred, const unsigned char* pseudo, const unsigned char* aff_labels, int B, int nc,
                       int g_h, int g_w, int H, int W, int radius, int ignore_index, float w_seg, float w_diver, float* losses,
                       float* d_seg, float* d_attn_pred, void* workspace, void* stream);

/* The decoder head in training mode (SegFormerHead + DecoderTransformer + attn_pred, model/model_excel.py:60-76), exact fp32:
 *   forward_train keeps the activations in `workspace` and returns seg [B,nc,g,g] and attn_pred [B,P,P];
 *   backward (same all_feats / workspace) writes d loss / d parameter for every parameter into `grads`, a table with the layout
 *   of the weights whose pointers are WRITTEN (device memory of each parameter's shape); d_attn_pred may be NULL.
 *   dropout_p / dropout_seed: the head's Dropout2d (segformer_head.py:66,75; the reference trains with p = 0.1) as a counter-based
 *   channel mask, a pure function of (seed, image, channel): pass the same pair to forward_train and backward.  g*g % 4 == 0.
 * excel_adamw_step: torch.optim.AdamW update of one tensor (decoupled weight decay, bias correction with `step` >= 1), the
 * arithmetic under utils/optimizer.py's PolyWarmupAdamW; the learning-rate schedule stays on the host. */
size_t excel_decoder_train_workspace_bytes(excel_decoder_t h, int B, int g);
int excel_decoder_forward_train(excel_decoder_t h, const float* all_feats, int B, int g, void* workspace, size_t workspace_bytes,
                                float* seg_out, float* attn_pred_out, float dropout_p, unsigned dropout_seed, void* stream);
/* attn_fts [B,E,g,g] of the last excel_decoder_forward_train on `workspace`: the fused features AFTER the head's Dropout2d, what the
 * reference's training loop clones as the LVC cue (scripts/train_voc.py:186-189: fts_diver = attn_fts.clone().detach()). */
int excel_decoder_train_attn_fts(excel_decoder_t h, int B, int g, const void* workspace, size_t workspace_bytes, float* attn_fts_out,
                                 void* stream);
int excel_decoder_backward(excel_decoder_t h, const float* all_feats, int B, int g, void* workspace, size_t workspace_bytes,
                           const float* d_seg, const float* d_attn_pred, const excel_decoder_weights* grads, float dropout_p,
                           unsigned dropout_seed, void* stream);
int excel_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, void* stream);

/* lam_to_label (utils/camutils.py:123-145): cam [B,F,H,W], cls_label [B,F] -> valid_cam = cls*cam (optional) and label u8 [B,H,W] =
 * argmax+1, thresholded (ignore_mid: value <= high -> ignore_index, value <= low -> 0; else value <= bkg -> 0); img_box [B,4] int32
 * (y0,y1,x0,x1; optional): pixels outside the box -> ignore_index. */
int excel_lam_to_label(const float* cam, const float* cls_label, const int32_t* img_box, int B, int F, int H, int W, float bkg_thre,
                       float high_thre, float low_thre, int ignore_mid, int ignore_index, float* valid_cam, unsigned char* label,
                       void* stream);

/* transforms.normalize_img + the HWC->CHW transpose of the dataset (datasets/transforms.py; datasets/voc.py:115-116):
 * hwc [B,H,W,3] uint8 (decoded image) -> out [B,3,H,W] f32 = (u8 - mean[c]) / std[c], double intermediate like numpy.  mean3/std3: HOST doubles. */
int excel_normalize_img_u8(const unsigned char* hwc, int B, int H, int W, const double* mean3, const double* std3, float* out, void* stream);

/* denormalize_img / denormalize_img2 (utils/imutils.py:11-25): img [B,3,H,W] f32 (normalised) -> v = img*std[c] + mean[c]
 * truncated to uint8 (out_u8, optional) and/or that value / 255 as float (out_f32, optional).  mean3/std3: HOST pointers to 3 floats. */
int excel_denormalize_img(const float* img, int B, int H, int W, const float* mean3, const float* std3, unsigned char* out_u8,
                          float* out_f32, void* stream);

/* Multi-scale / flip fuse of the segmentation logits (tools/infer_seg_voc.py:66-82): segs [2B,nc,h,w] of one scale (second
 * half from the x-flipped inputs) -> bilinear (align_corners=False) to (H,W) -> flip_mean ? (seg + flip_x(seg_flipped))/2 : seg
 * (scale 1.0 uses the un-flipped half alone, :69) -> acc [B,nc,H,W] = ((init ? 0 : acc) + .) * scale (mean over scales: pass
 * 1/n_scales with the last scale, 1 otherwise). */
int excel_seg_scale_accumulate(const float* segs, float* acc, int B, int nc, int h, int w, int H, int W, int flip_mean, int init,
                               float scale, void* stream);

/* ------------------------------------------------------------------ patch-text CAM */

/* clip_feature_surgery (clip/clip.py:288-310, redundant_feats=None) on normalised features:
 *   image_features [B,N,C], text [T,C] (unit rows) -> out_full [B,N,T] and/or the caller's slice
 *   out_slice [B,N-1,F] = out_full[:, 1:, :F] (model/model_excel.py:58).  workspace: B*N*ldT floats, ldT = T rounded up to 4. */
size_t excel_cam_workspace_bytes(int B, int N, int T);
int excel_clip_feature_surgery(const float* image_features, const float* text, int B, int N, int C, int T, int F,
                               float temperature, float* out_full, float* out_slice, void* workspace, void* stream);

/* The same attribute maps from the UN-normalised token features (x_raw of excel_vit_forward): token-axis L2 norm (clip/clip.py:353) +
 * similarity GEMM on the matrix core + the surgery epilogue (clip/clip.py:288-310) -- model/model_excel.py:57-58 back to back.  Three
 * launches, each over the whole chip: column sums of squares in image-aligned row blocks (fixed order: an image's maps do not depend
 * on its position in the batch), the similarity tiles (one wave per 32-token tile, >= 7 workgroups per image), and the min-max
 * normalisation over a two-stage (exact) min / max reduction.
 *   mode 1: bf16x3 (fp32 operands as bf16 hi+lo, 3 MFMAs per product), mode 2: f16x3 (IEEE-half planes), mode 0: exact fp32 MFMA.
 *   image_features [B,N,C] (optional): the normalised features generate_clip_fts returns.
 * T <= 128, C % 32 == 0, C <= 1024; x_raw, text, image_features and the workspace 16-byte aligned. */
size_t excel_patch_text_cam_workspace_bytes(int B, int N, int C, int T);
int excel_patch_text_cam(const float* x_raw, const float* text, int B, int N, int C, int T, int F, float temperature, int mode,
                         float* out_full, float* out_slice, float* image_features, void* workspace, void* stream);

/* ------------------------------------------------------------------ DenseCRF post-processing */

/* utils/dcrf.py:42-68 (class DenseCRF; :7-40 crf_inference / crf_inference_label use the same call with other parameters), driven by
 * tools/infer_lam.py:179-237: DenseCRF2D(W, H, C) + setUnaryEnergy + addPairwiseGaussian(sxy = pos_xy_std, compat = pos_w) +
 * addPairwiseBilateral(sxy = bi_xy_std, srgb = bi_rgb_std, rgbim, compat = bi_w) + inference(iters) of pydensecrf (DIAG kernels,
 * symmetric normalisation, Potts compatibilities): mean-field inference with permutohedral-lattice message passing, all on the device.
 *   rgb_hwc [H,W,3] uint8, prob [C,H,W]: probabilities (prob_is_energy = 0: unary = -log(clip(p, 1e-5, 1)), pydensecrf's
 *   unary_from_softmax) or the unary energies themselves (prob_is_energy = 1, e.g. unary_from_labels)  ->  q_out [C,H,W]. */
size_t excel_dcrf_workspace_bytes(int H, int W, int C);
int excel_dcrf_inference(const unsigned char* rgb_hwc, const float* prob, int prob_is_energy, int H, int W, int C, int iters, float pos_w,
                         float pos_xy_std, float bi_w, float bi_xy_std, float bi_rgb_std, float* q_out, void* workspace, void* stream);

/* ------------------------------------------------------------------ affinity random walk */

/* mean over layers of attn[l, 1:, 1:] for one stacked tensor [Lw,B,N,N] -> [B,P,P] (utils/affutils.py:180,197). */
int excel_attn_layer_mean(const float* attn, int Lw, int B, int N, int first_layer, int n_layers, float* w_aff, void* stream);

/* compute_trans_mat (utils/affutils.py:8-24): 3x(col,row) normalise, symmetrise, square (batched fp32 MFMA GEMM).
 * workspace: (2*B*P*P + B*P) floats. */
/* seg_attn branch of refine_cams_with_aff (utils/affutils.py:182-195): of the n_layers per-layer maps
 * attn[first_layer .. first_layer+n_layers) ([Lw,B,N,N] stacked, [1:,1:] used) keep those whose total difference to
 * seg_attn [B,P,P] is <= the mean difference, average them (/ (count + 1e-5)) and gate by seg_attn -> w_out [B,P,P]. */
size_t excel_attn_select_workspace_bytes(int B, int n_layers);
int excel_attn_select_mean(const float* attn, int Lw, int B, int N, int first_layer, int n_layers, const float* seg_attn,
                           float* w_out, void* workspace, void* stream);

size_t excel_trans_mat_workspace_bytes(int B, int P);
int excel_compute_trans_mat(const float* w_aff, int B, int P, float* trans_out, void* workspace, void* stream);

/* present-class compaction of one-hot labels [B,F] -> cls_idx [B,Smax] (-1 padded), ncls [B], and optionally
 * nchan [B] = min(ncls,Smax)+1 (channels incl. background)  (cls_lst = torch.where(cls_label)[0], utils/affutils.py:203). */
int excel_cls_compact(const float* onehot, int B, int F, int Smax, int32_t* cls_idx, int32_t* ncls, int32_t* nchan, void* stream);

/* scoremap2bbox + box mask (utils/affutils.py:26-53, :208-214) for every (image, present class):
 *   attr [B,P,F] -> v [B,Smax,P] = mask .* attr[:, :, cls]; mask_out [B,Smax,P] u8 optional. */
int excel_scoremap_box_mask(const float* attr, const int32_t* cls_idx, const int32_t* ncls, int B, int g, int F, int Smax,
                            double caa_thre, float* v_out, uint8_t* mask_out, void* stream);

/* refine_cams_with_aff, batched and fused (utils/affutils.py:177-223, seg_attn=None):
 *   refined[b,s,:] = T2 (mask_s .* attr[:, cls_s]) with T2 = Tsym.Tsym applied as two mat-vecs.
 *   workspace: excel_refine_workspace_bytes.  refined [B,Smax,P]. */
size_t excel_refine_workspace_bytes(int B, int P, int Smax);
int excel_refine_cams_with_aff(const float* attr, const float* w_aff, const int32_t* cls_idx, const int32_t* ncls,
                               int B, int g, int F, int Smax, double caa_thre, float* refined, void* workspace, void* stream);

/* generate_cam_label/scale_cam_image + background channel (utils/affutils.py:55-78, :164-166):
 *   refined [B,Smax,P] -> cams [B,Smax+1,H,W]: channel 0 = 1 - max_c, channels 1..ncls[b] = min-max normalised,
 *   bilinearly (cv2.resize INTER_LINEAR rule) up-sampled maps.  workspace: B*Smax*P floats.
 *   flags: EXCEL_CAMS_ZERO_UNUSED = also write zeros to channels ncls[b]+1..Smax (nothing on the path reads them). */
#define EXCEL_CAMS_ZERO_UNUSED 1
int excel_cam_upsample_bkg(const float* refined, const int32_t* ncls, int B, int g, int Smax, int H, int W, float* cams,
                           void* workspace, int flags, void* stream);

/* ------------------------------------------------------------------ PAR + labels + metric */

/* PAR.forward (utils/PAR.py:64-92): imgs [B,3,h,w], masks [B,Cmax,H,W] -> out [B,Cmax,H,W] after n_iter steps.
 * nchan (optional) [B]: only channels < nchan[b] of image b are refined (ragged class counts).
 * workspace: excel_par_workspace_bytes (aff planes + ping-pong + resized guide).
 * Affinities: by default the 8*ndil weights of a pixel are recomputed inside every Jacobi step from the guide image and 5 per-pixel
 * statistics (20 B/pixel instead of 192 B/pixel of HBM traffic per step) where the tiled kernel applies (dilations [1,2,4,8,12,24] -
 * the only set the reference uses: tools/infer_lam.py:168 - W % 4 == 0, 16-byte aligned buffers); otherwise, or with
 * flags & EXCEL_PAR_STREAM_AFFINITIES, they are streamed as planes.  Same arithmetic in the same order: the outputs are bit-identical. */
#define EXCEL_PAR_STREAM_AFFINITIES 1
size_t excel_par_workspace_bytes(int B, int Cmax, int H, int W, int ndil);
int excel_par_forward(const float* imgs, int h, int w, const float* masks, const int32_t* nchan, int B, int Cmax, int H, int W,
                      const int32_t* dilations /*host*/, int ndil, int n_iter, float w1, float w2, float* out,
                      void* workspace, int flags, void* stream);

/* refined.argmax(1) -> valid_key lookup (utils/affutils.py:86-87, :168).  cls_idx may be NULL (identity keys). */
int excel_argmax_label(const float* cams, const int32_t* nchan, const int32_t* cls_idx, int B, int Smax, int Cmax,
                       long long HW, uint8_t* labels_u8, int64_t* labels_i64, void* stream);

/* _fast_hist accumulated on device (utils/evaluate.py:9-20): hist[nc*gt+pred] += 1 for gt < nc; hist is int64 [nc*nc]. */
int excel_confusion_accumulate(const uint8_t* gt, const uint8_t* pred, long long n, int num_classes, int64_t* hist, void* stream);

/* ------------------------------------------------------------------ ragged batches: B images of DIFFERENT label sizes in one launch
 * The reference evaluates at every image's own label size (tools/infer_lam.py:74,94: the input is resized to S x S, everything after
 * the random walk runs at labels.shape[-2:]) with batch 1; these entry points run the size-dependent half of the path (input resize,
 * CAM up-sampling, PAR, arg-max) for a whole batch at once.  The ViT / CAM / random-walk half is size-uniform (S x S) already.
 *
 * Layout ("pitched planes"): image b has H_b x W_b pixels; rows are padded to Wp_b = W_b rounded up to 4 floats, a plane is
 * H_b * Wp_b floats; image b of a K-plane fp32 tensor starts at element K * poff_b (poff_b = sum_{i<b} H_i * Wp_i) with its planes
 * back to back.  uint8 maps (decoded HWC images, ground truth, labels) are TIGHT: image b at pixel loff_b = sum_{i<b} H_i * W_i, so
 * excel_confusion_accumulate runs over the flat label arrays unchanged.  Work is cut into 64 x 16 pixel tiles, numbered image-major.
 * excel_ragged_plan (a HOST function: no device work) fills `info` and, when `table` != NULL, the int32 table the kernels read:
 *   (B+1) records of 8 ints {H, W, poff, tile_off, loff, 0, 0, 0} (record B = totals), then the image index of every tile;
 * the caller copies `table` (info->table_ints int32) to the device.  Offsets are 32-bit: total padded pixels < 2^31. */
typedef struct {
    int32_t B, total_tiles;
    int64_t total_pix;         /* sum H_b * Wp_b: elements of a one-plane pitched tensor */
    int64_t total_label_pix;   /* sum H_b * W_b:  elements of a tight u8 map */
    int64_t max_plane_pix;     /* max H_b * Wp_b */
    int64_t table_ints;
} excel_ragged_info;
int excel_ragged_plan(const int32_t* hw /*host [B,2] = (H_b, W_b)*/, int B, excel_ragged_info* info /*host*/, int32_t* table /*host or NULL*/);

/* datasets/transforms.normalize_img + HWC->CHW + the harness' input resize (tools/infer_lam.py:74: F.interpolate bilinear,
 * align_corners=False, no antialias) for decoded images of different sizes: hwc = the uint8 [H_b,W_b,3] images back to back
 * (image b at byte 3 * loff_b) -> out [B,3,S,S] f32.  Same operations (same bits) as excel_normalize_img_u8 + excel_bilinear_resize. */
int excel_normalize_resize_u8_ragged(const uint8_t* hwc, const int32_t* table, int B, int S, const double* mean3 /*host*/,
                                     const double* std3 /*host*/, float* out, void* stream);

/* excel_cam_upsample_bkg for a ragged batch: refined [B,Smax,P] -> cams = (Smax+1) pitched planes per image. workspace: B*Smax*P floats. */
int excel_cam_upsample_bkg_ragged(const float* refined, const int32_t* ncls, const int32_t* table, const excel_ragged_info* info, int g,
                                  int Smax, float* cams, void* workspace, int flags, void* stream);

/* PAR.forward for a ragged batch: imgs [B,3,h,w] (the uniform network input; resized per image with align_corners=True like
 * PAR.py:67), masks / out = Cmax pitched planes per image.
 * Narrower contract than excel_par_forward (only the recomputing tile kernel exists for ragged batches; there is no streamed-affinity
 * fallback): returns EXCEL_ERR_ARG (-1) unless
 *   - n_iter >= 1            (excel_par_forward copies masks to out for n_iter == 0; here that is an error),
 *   - dilations == [1,2,4,8,12,24] (what the reference always uses, PAR.py callers),
 *   - every buffer is 16-byte aligned and max(Cmax,5) * H_b * Wp_b * 4 < 2^31 for every image (Wp_b = W_b rounded up to 4).
 * The pad columns (W_b <= x < Wp_b) of `out` and its planes c >= nchan[b] are UNDEFINED on return (masks' pad columns are never read
 * as neighbours); consumers clamp to W_b - 1 and nchan[b] (excel_argmax_label_ragged does). */
size_t excel_par_ragged_workspace_bytes(long long total_pix, int Cmax);
int excel_par_forward_ragged(const float* imgs, int h, int w, const float* masks, const int32_t* nchan, const int32_t* table,
                             const excel_ragged_info* info, int Cmax, const int32_t* dilations /*host*/, int ndil, int n_iter, float w1,
                             float w2, float* out, void* workspace, void* stream);

/* excel_argmax_label for a ragged batch: cams = Cmax pitched planes per image -> tight uint8 labels (info->total_label_pix). */
int excel_argmax_label_ragged(const float* cams, const int32_t* nchan, const int32_t* cls_idx, const int32_t* table,
                              const excel_ragged_info* info, int Smax, int Cmax, uint8_t* labels_u8, void* stream);

/* ------------------------------------------------------------------ one-time / auxiliary */

/* attr_aggregate (model/load_attr.py:86-119): text [T,C] (F fg rows first), bank [C,K] -> text_attr [C,T], columns unit-norm.
 * topK as in the reference (0.9): the lowest int((1-topK)*K) logits per fg row are dropped before the softmax. */
int excel_attr_aggregate(const float* text, const float* bank, int F, int T, int C, int K, double topK, float* out, void* stream);

/* F.interpolate(mode='bilinear', align_corners=0|1) on `planes` images of h x w -> H x W
 * (tools/infer_lam.py:74 input resize; utils/camutils.py:41,54 multi-scale CAM resize; utils/PAR.py:67). */
int excel_bilinear_resize(const float* in, float* out, long long planes, int h, int w, int H, int W, int align_corners, void* stream);

/* positional_embedding [1+side^2, D] -> [1+g^2, D] (clip/clip_surgery_model.py:407-414 and :426-435). */
int excel_pos_embed_resize(const float* pos, int side, int g, int D, float* out, void* stream);

/* flip-TTA fuse of cure_attr_map_flip (utils/camutils.py:21-26): attr [2B,P,F] -> out [B,P,F]. */
int excel_flip_max_normalize(const float* attr, float* out, int B, int g, int F, void* stream);

/* multi-scale LAM fuse (utils/camutils.py:41-61, multi_scale_lam2 in its evident intent): for one scale, maps [2B,P,F]
 * (second half from horizontally flipped inputs) are bilinearly resized (align_corners=False) to (H,W), flip-maxed and
 * accumulated into acc [B,F,H,W] (init=1: overwrite).  excel_plane_minmax_normalize then applies
 * lam -= min_hw; lam /= max_hw + 1e-5 per (b,f) plane. */
int excel_lam_scale_accumulate(const float* maps, float* acc, int B, int g, int F, int H, int W, int init, void* stream);
int excel_plane_minmax_normalize(float* lam, long long planes, long long HW, void* stream);

/* ------------------------------------------------------------------ live per-kernel timing (bench.py) */

/* When enabled, every kernel launch is bracketed by hipEvents on its launch stream, grouped by category.
 * excel_prof_collect synchronises, fills ms[c] (summed elapsed), launches[c], work[c] (algorithmic FLOPs for the GEMM
 * categories, 0 otherwise) for c < excel_prof_num_categories(), and clears the log. */
int excel_prof_enable(int on);
int excel_prof_set_mask(unsigned long long category_mask);   /* bit c: bracket category c (default all) */
int excel_prof_set_sampling(int every);                      /* bracket every n-th launch of a category (default 1) */
int excel_prof_num_categories(void);
const char* excel_prof_category_name(int cat);
int excel_prof_collect(double* ms /*host*/, long long* launches /*host*/, double* work /*host*/);

#ifdef __cplusplus
}
#endif
#endif
