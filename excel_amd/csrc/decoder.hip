// Decoder head inference: SegFormerHead fuse (model/segformer_head.py:47-77) + DecoderTransformer
// (model/decoder/TransDecoder.py:62-124) over the ViT's per-block features.  A small model (width 256, 3 layers, 8 heads of
// 32): orchestration of the exact-fp32 matrix-core GEMM with a row softmax in between; scores are materialised
// ([B*heads, P, P] fp32, 0.6 GB at B=32, 448^2) -- at ~7 % of the ViT's flops this head is not worth a fused kernel yet.
#include "decoder_internal.h"

// in-place softmax over rows of length P stored with pitch Pp (pad columns are zeroed: they are the K tail of P.V)
// causal: row q (= row index modulo P) attends to keys 0..q only (the additive -inf mask of build_attention_mask,
// clip/clip_surgery_model.py:537-543)
__global__ __launch_bounds__(256) void dec_row_softmax_kernel(float* __restrict__ s, long long rows, int P, int Pp, int causal) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* r = s + row * Pp;
    const int n = causal ? (int)(row % P) + 1 : P;
    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, r[i]);
    m = wave_max(m);
    float sum = 0.f;
    for (int i = lane; i < n; i += 64) sum += expf(r[i] - m);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int i = lane; i < Pp; i += 64) r[i] = i < n ? expf(r[i] - m) * inv : 0.f;
}

// [B, R, Cc] (pitch ld) -> [B, Cc, Rp]   (token-major -> channel-major maps; columns r >= R of the output are zero-filled)
__global__ __launch_bounds__(256) void dec_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc, int ld, int Rp) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < Cc) ? in[((long long)b * R + r) * ld + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < Cc && r < Rp) out[((long long)b * Cc + c) * Rp + r] = tile[tx][j];
    }
}

int excel_launch_dec_transpose(const float* in, float* out, int B, int R, int Cc, int ld, int Rp, hipStream_t st) {
    hipLaunchKernelGGL(dec_transpose_kernel, dim3(cdiv(Rp, 32), cdiv(Cc, 32), B), dim3(256), 0, st, in, out, R, Cc, ld, Rp);
    EXCEL_CHECK_LAUNCH("dec_transpose");
    return EXCEL_OK;
}

int excel_launch_dec_softmax(float* s, long long rows, int P, int Pp, int causal, hipStream_t st) {
    hipLaunchKernelGGL(dec_row_softmax_kernel, dim3((unsigned)cdivl(rows, 4)), dim3(256), 0, st, s, rows, P, Pp, causal);
    EXCEL_CHECK_LAUNCH("dec_softmax");
    return EXCEL_OK;
}

struct DecWs {
    float *h1, *cat, *x, *y, *qkv, *s, *ao, *hb, *segt;
    int Pp, ncp;
    size_t total;
};

static DecWs dec_ws_layout(const excel_decoder_config& c, int B, int g, char* base) {
    DecWs w;
    const int P = g * g, E = c.embed;
    const size_t M = (size_t)B * P;
    w.Pp = (P + 3) / 4 * 4;
    w.ncp = (c.num_classes + 3) / 4 * 4;
    size_t off = 0;
    auto take = [&](size_t floats) { float* p = (float*)(base + off); off += align_up(floats * sizeof(float), 256); return p; };
    w.h1 = take(M * E);
    w.cat = take(M * (size_t)c.vit_layers * E);
    w.x = take(M * E);
    w.y = take(M * E);
    w.qkv = take(M * 3 * E);
    w.s = take((size_t)B * c.heads * P * w.Pp);
    w.ao = take(M * E);
    w.hb = take(M * 4 * E);
    w.segt = take(M * w.ncp);
    w.total = off;
    return w;
}

// Stack of pre-LN residual blocks (nn.MultiheadAttention + QuickGELU MLP): x [B*P, E] updated in place.
// Shared by the decoder head (TransDecoder.py:62-84) and the CLIP text tower (clip_surgery_model.py:285-337 with the causal mask).
static int preln_blocks(const excel_decoder_block_weights* blocks, int n_layers, float* x, float* y, float* qkv, float* sbuf, float* ao,
                        float* hb, int B, int P, int Pp, int E, int H, int causal, hipStream_t st) {
    const int hd = E / H, M = B * P;
    const float scale = 1.f / sqrtf((float)hd);
    for (int l = 0; l < n_layers; ++l) {
        const excel_decoder_block_weights& bw = blocks[l];
        TRYD(excel_launch_layernorm(x, nullptr, 1, bw.ln1_w, bw.ln1_b, y, M, E, 1e-5f, st));
        GemmArgs q = ga0(y, bw.in_proj_w, qkv, bw.in_proj_b, nullptr, M, 3 * E, E, E, E, 3 * E, 0, GEMM_ACT_NONE);
        q.out_mode = GEMM_OUT_QKV_HEADMAJOR; q.tokN = P; q.heads = H; q.hd = hd;                 // -> [B,3,H,P,hd]
        TRYD(excel_launch_gemm(q, true, 1, st));
        // scores[b,h] = scale * q k^T   (batched NT over (b,h))
        GemmArgs sc = ga0(qkv, qkv + (size_t)H * P * hd, sbuf, nullptr, nullptr, P, P, hd, hd, hd, Pp, 0, GEMM_ACT_NONE);
        sc.alpha = scale; sc.zdiv = H;
        sc.sA = sc.sB = (long long)3 * H * P * hd; sc.sA2 = sc.sB2 = (long long)P * hd;
        sc.sC = (long long)H * P * Pp; sc.sC2 = (long long)P * Pp;
        TRYD(excel_launch_gemm(sc, true, B * H, st));
        hipLaunchKernelGGL(dec_row_softmax_kernel, dim3((unsigned)cdivl((long long)B * H * P, 4)), dim3(256), 0, st, sbuf, (long long)B * H * P,
                           P, Pp, causal);
        EXCEL_CHECK_LAUNCH("preln_blocks/softmax");
        // out[b, :, h*hd:(h+1)*hd] = P[b,h] . v[b,h]   (batched NN, heads merged through the column offset)
        GemmArgs pv = ga0(sbuf, qkv + (size_t)2 * H * P * hd, ao, nullptr, nullptr, P, hd, P, Pp, hd, E, 0, GEMM_ACT_NONE);
        pv.Kld = Pp; pv.zdiv = H;
        pv.sA = (long long)H * P * Pp; pv.sA2 = (long long)P * Pp;
        pv.sB = (long long)3 * H * P * hd; pv.sB2 = (long long)P * hd;
        pv.sC = (long long)P * E; pv.sC2 = hd;
        TRYD(excel_launch_gemm(pv, false, B * H, st));
        GemmArgs op = ga0(ao, bw.out_proj_w, x, bw.out_proj_b, x, M, E, E, E, E, E, E, GEMM_ACT_NONE);       // x += out_proj(.)
        TRYD(excel_launch_gemm(op, true, 1, st));
        TRYD(excel_launch_layernorm(x, nullptr, 1, bw.ln2_w, bw.ln2_b, y, M, E, 1e-5f, st));
        GemmArgs f1 = ga0(y, bw.fc1_w, hb, bw.fc1_b, nullptr, M, 4 * E, E, E, E, 4 * E, 0, GEMM_ACT_QUICKGELU);
        TRYD(excel_launch_gemm(f1, true, 1, st));
        GemmArgs f2 = ga0(hb, bw.fc2_w, x, bw.fc2_b, x, M, E, 4 * E, 4 * E, 4 * E, E, E, GEMM_ACT_NONE);      // x += mlp(ln_2(x))
        TRYD(excel_launch_gemm(f2, true, 1, st));
    }
    return EXCEL_OK;
}

extern "C" int excel_decoder_create(const excel_decoder_config* cfg, const excel_decoder_weights* w, excel_decoder_t* out) {
    EXCEL_CHECK_ARG(cfg && w && out && w->fuse && w->blocks, "excel_decoder_create: null argument");
    EXCEL_CHECK_ARG(cfg->vit_layers >= 1 && cfg->dec_layers >= 0 && cfg->heads >= 1 && cfg->embed % cfg->heads == 0 &&
                        (cfg->embed / cfg->heads) % 4 == 0 && cfg->vit_width % 4 == 0 && cfg->num_classes >= 1,
                    "excel_decoder_create: embed must split into heads of a multiple of 4, vit_width %% 4 == 0");
    excel_decoder* h = new excel_decoder();
    h->cfg = *cfg;
    h->w = *w;
    h->fuse.assign(w->fuse, w->fuse + cfg->vit_layers);
    h->blocks.assign(w->blocks, w->blocks + cfg->dec_layers);
    h->w.fuse = h->fuse.data();
    h->w.blocks = h->blocks.data();
    *out = h;
    return EXCEL_OK;
}

extern "C" void excel_decoder_destroy(excel_decoder_t h) { delete h; }

extern "C" size_t excel_decoder_workspace_bytes(excel_decoder_t h, int B, int g) {
    if (!h || B <= 0 || g <= 0) return 0;
    return dec_ws_layout(h->cfg, B, g, nullptr).total;
}


extern "C" int excel_decoder_forward(excel_decoder_t h, const float* all_feats, int B, int g, void* workspace, size_t workspace_bytes,
                                     float* attn_fts_out, float* seg_out, void* stream) {
    EXCEL_CHECK_ARG(h && all_feats && workspace && B > 0 && g > 0, "excel_decoder_forward: bad argument");
    const excel_decoder_config& c = h->cfg;
    const int P = g * g, N = P + 1, D = c.vit_width, E = c.embed, L = c.vit_layers, H = c.heads, hd = E / H, nc = c.num_classes;
    const int M = B * P;
    hipStream_t st = (hipStream_t)stream;
    DecWs ws = dec_ws_layout(c, B, g, (char*)workspace);
    EXCEL_CHECK_ARG(workspace_bytes >= ws.total, "excel_decoder_forward: workspace too small (%zu < %zu)", workspace_bytes, ws.total);

    // ---- SegFormerHead: per ViT layer Linear -> ReLU -> Linear on the patch tokens (cls row skipped), channel concat, 1x1 fuse
    for (int l = 0; l < L; ++l) {
        const excel_fuse_layer_weights& fw = h->fuse[l];
        GemmArgs a = ga0(all_feats + ((size_t)l * B * N + 1) * D, fw.proj_w, ws.h1, fw.proj_b, nullptr, P, E, D, D, D, E, 0, GEMM_ACT_RELU);
        a.sA = (long long)N * D; a.sC = (long long)P * E;                       // one image per batch entry: rows 1..P of [N,D]
        TRYD(excel_launch_gemm(a, true, B, st));                                // segformer_head.py:23-24
        GemmArgs b2 = ga0(ws.h1, fw.proj2_w, ws.cat + (size_t)l * E, fw.proj2_b, nullptr, M, E, E, E, E, L * E, 0, GEMM_ACT_NONE);
        TRYD(excel_launch_gemm(b2, true, 1, st));                               // :25, written at channel offset l*E (:73)
    }
    GemmArgs fz = ga0(ws.cat, h->w.fuse_w, ws.x, h->w.fuse_b, nullptr, M, E, L * E, L * E, L * E, E, 0, GEMM_ACT_NONE);
    TRYD(excel_launch_gemm(fz, true, 1, st));                                   // :74 (dropout inactive in eval)
    if (attn_fts_out) {
        hipLaunchKernelGGL(dec_transpose_kernel, dim3(cdiv(P, 32), cdiv(E, 32), B), dim3(256), 0, st, ws.x, attn_fts_out, P, E, E, P);
        EXCEL_CHECK_LAUNCH("decoder/attn_fts");
    }
    if (!seg_out) return EXCEL_OK;

    // ---- DecoderTransformer: pre-LN residual blocks (TransDecoder.py:78-83), tokens [B*P, E]
    TRYD(preln_blocks(h->blocks.data(), c.dec_layers, ws.x, ws.y, ws.qkv, ws.s, ws.ao, ws.hb, B, P, ws.Pp, E, H, 0, st));
    GemmArgs lp = ga0(ws.x, h->w.pred_w, ws.segt, h->w.pred_b, nullptr, M, nc, E, E, E, ws.ncp, 0, GEMM_ACT_NONE);   // linear_pred (:122)
    TRYD(excel_launch_gemm(lp, true, 1, st));
    hipLaunchKernelGGL(dec_transpose_kernel, dim3(cdiv(P, 32), cdiv(nc, 32), B), dim3(256), 0, st, ws.segt, seg_out, P, nc, ws.ncp, P);
    EXCEL_CHECK_LAUNCH("decoder/seg");
    return EXCEL_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// CLIP text tower (one-time text-bank builder): encode_text (clip/clip_surgery_model.py:551-564) =
// token embedding + positional embedding -> causal pre-LN transformer -> ln_final -> row of the EOT token (the largest
// token id of the sequence) @ text_projection.
struct excel_text {
    excel_text_config cfg;
    excel_text_weights w;
    std::vector<excel_decoder_block_weights> blocks;
};

__global__ __launch_bounds__(256) void text_embed_kernel(const int* __restrict__ tok, const float* __restrict__ emb, const float* __restrict__ pos,
                                                         float* __restrict__ x, int ctx, int E, int vocab) {
    const int row = blockIdx.x;                     // b * ctx + t
    const int t = row % ctx;
    const int id = min(max(tok[row], 0), vocab - 1);
    for (int i = threadIdx.x; i < E; i += 256) x[(long long)row * E + i] = emb[(long long)id * E + i] + pos[(long long)t * E + i];
}

// eot[b] = first arg-max of tok[b, :] (torch.argmax), rows[b] = xln[b, eot[b], :]
__global__ __launch_bounds__(64) void text_eot_gather_kernel(const int* __restrict__ tok, const float* __restrict__ xln, float* __restrict__ rows,
                                                             int ctx, int E) {
    const int b = blockIdx.x;
    int best = tok[(long long)b * ctx], bi = 0;
    for (int t = 1; t < ctx; ++t) { const int v = tok[(long long)b * ctx + t]; if (v > best) { best = v; bi = t; } }
    for (int i = threadIdx.x; i < E; i += 64) rows[(long long)b * E + i] = xln[((long long)b * ctx + bi) * E + i];
}

// encode_text_with_prompt_ensemble (clip/clip.py:262-266): rows /= ||row||; mean over rows; /= ||mean||.  One workgroup.
__global__ __launch_bounds__(256) void prompt_ensemble_kernel(const float* __restrict__ emb, float* __restrict__ out, int n, int E) {
    extern __shared__ float sm[];                   // [n] inverse norms, then reduction scratch
    float* inv = sm;
    float* red = sm + n;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < n; r += 4) {
        float ss = 0.f;
        for (int i = lane; i < E; i += 64) { const float v = emb[(long long)r * E + i]; ss = fmaf(v, v, ss); }
        ss = wave_sum(ss);
        if (lane == 0) inv[r] = 1.f / sqrtf(ss);
    }
    __syncthreads();
    float part = 0.f;
    for (int i = threadIdx.x; i < E; i += 256) {
        float acc = 0.f;
        for (int r = 0; r < n; ++r) acc += emb[(long long)r * E + i] * inv[r];
        acc /= (float)n;
        out[i] = acc;
        part = fmaf(acc, acc, part);
    }
    part = wave_sum(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    for (int i = threadIdx.x; i < E; i += 256) out[i] = out[i] / nrm;
}

struct TextWs { float *x, *y, *qkv, *s, *ao, *hb, *rows; int Pp; size_t total; };
static TextWs text_ws_layout(const excel_text_config& c, int B, char* base) {
    TextWs w;
    const int P = c.context_length, E = c.width;
    const size_t M = (size_t)B * P;
    w.Pp = (P + 3) / 4 * 4;
    size_t off = 0;
    auto take = [&](size_t floats) { float* p = (float*)(base + off); off += align_up(floats * sizeof(float), 256); return p; };
    w.x = take(M * E); w.y = take(M * E); w.qkv = take(M * 3 * E); w.s = take((size_t)B * c.heads * P * w.Pp);
    w.ao = take(M * E); w.hb = take(M * 4 * E); w.rows = take((size_t)B * E);
    w.total = off;
    return w;
}

extern "C" int excel_text_create(const excel_text_config* cfg, const excel_text_weights* w, excel_text_t* out) {
    EXCEL_CHECK_ARG(cfg && w && out && w->blocks && w->token_embedding && w->positional_embedding && w->text_projection, "excel_text_create: null argument");
    EXCEL_CHECK_ARG(cfg->width % cfg->heads == 0 && (cfg->width / cfg->heads) % 4 == 0 && cfg->embed_dim % 4 == 0 && cfg->context_length >= 1,
                    "excel_text_create: width must split into heads of a multiple of 4, embed_dim %% 4 == 0");
    excel_text* h = new excel_text();
    h->cfg = *cfg;
    h->w = *w;
    h->blocks.assign(w->blocks, w->blocks + cfg->layers);
    h->w.blocks = h->blocks.data();
    *out = h;
    return EXCEL_OK;
}

extern "C" void excel_text_destroy(excel_text_t h) { delete h; }

extern "C" size_t excel_text_workspace_bytes(excel_text_t h, int B) { return (h && B > 0) ? text_ws_layout(h->cfg, B, nullptr).total : 0; }

extern "C" int excel_text_encode(excel_text_t h, const int32_t* tokens, int B, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    EXCEL_CHECK_ARG(h && tokens && out && workspace && B > 0, "excel_text_encode: bad argument");
    const excel_text_config& c = h->cfg;
    const int P = c.context_length, E = c.width, M = B * P;
    hipStream_t st = (hipStream_t)stream;
    TextWs ws = text_ws_layout(c, B, (char*)workspace);
    EXCEL_CHECK_ARG(workspace_bytes >= ws.total, "excel_text_encode: workspace too small (%zu < %zu)", workspace_bytes, ws.total);
    hipLaunchKernelGGL(text_embed_kernel, dim3(M), dim3(256), 0, st, tokens, h->w.token_embedding, h->w.positional_embedding, ws.x, P, E, c.vocab_size);
    EXCEL_CHECK_LAUNCH("text/embed");                                                                                // :552-554
    TRYD(preln_blocks(h->blocks.data(), c.layers, ws.x, ws.y, ws.qkv, ws.s, ws.ao, ws.hb, B, P, ws.Pp, E, c.heads, 1, st));   // :556
    TRYD(excel_launch_layernorm(ws.x, nullptr, 1, h->w.ln_final_w, h->w.ln_final_b, ws.y, M, E, 1e-5f, st));       // :558
    hipLaunchKernelGGL(text_eot_gather_kernel, dim3(B), dim3(64), 0, st, tokens, ws.y, ws.rows, P, E);              // :562
    EXCEL_CHECK_LAUNCH("text/eot");
    // rows [B,E] @ text_projection [E, embed_dim]   (NN GEMM)
    GemmArgs pj = ga0(ws.rows, h->w.text_projection, out, nullptr, nullptr, B, c.embed_dim, E, E, c.embed_dim, c.embed_dim, 0, GEMM_ACT_NONE);
    return excel_launch_gemm(pj, false, 1, st);
}

extern "C" int excel_prompt_ensemble(const float* emb, int n, int E, float* out, void* stream) {
    EXCEL_CHECK_ARG(emb && out && n >= 1 && n <= 4096 && E >= 1, "excel_prompt_ensemble: bad argument");
    hipLaunchKernelGGL(prompt_ensemble_kernel, dim3(1), dim3(256), (n + 4) * sizeof(float), (hipStream_t)stream, emb, out, n, E);
    EXCEL_CHECK_LAUNCH("prompt_ensemble");
    return EXCEL_OK;
}
