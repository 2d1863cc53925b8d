/*
 * lemon_hip.h -- C ABI of liblemon_hip.so, the MI355X (gfx950) implementation of
 * LEMoN's kNN + multimodal-neighbour scoring hot path.
 *
 * The reference (MLforHealth/LEMoN, 100% Python) has no FFI layer; the seam this
 * library replaces is the object protocol run_lemon.py uses.  Each entry point
 * cites the reference call site it stands in for (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - every pointer named *_dev is a DEVICE pointer (hipMalloc / torch data_ptr());
 *     all matrices are row-major, float32 unless stated, int64 labels like faiss;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     call only enqueues work on that stream (no host synchronisation) unless its
 *     comment says otherwise;
 *   - return value: 0 = ok, <0 = error (LEMON_E_*); lemon_last_error() returns a
 *     thread-local description of the last failure;
 *   - caller owns every output buffer; the library owns only what is inside a
 *     lemon_index handle; one handle per thread.  The ONLY process-wide state is
 *     lemon_linear_f32's hipBLASLt handle + workspace + solution cache (one set per
 *     device, mutex-guarded; see its comment) and the thread-local error string.
 *   - numeric contract ("chain" numerics): dot(a,b) is the float32 fmaf chain in
 *     ascending k starting from +0 (bit-for-bit what v_mfma_f32_32x32x2_f32
 *     accumulates); ties in every top-k are broken towards the lower index.
 */
#ifndef LEMON_HIP_H
#define LEMON_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LEMON_METRIC_IP 0 /* faiss.IndexFlatIP, --dist_type cosine    (run_lemon.py:166-169) */
#define LEMON_METRIC_L2 1 /* faiss.IndexFlatL2, --dist_type euclidean (run_lemon.py:170-173) */

#define LEMON_OK 0
#define LEMON_E_INVALID (-1) /* bad argument (null pointer, d mismatch, k out of range) */
#define LEMON_E_HIP (-2)     /* a HIP runtime call failed; see lemon_last_error()        */
#define LEMON_E_NOMEM (-3)   /* device allocation failed                                 */

#define LEMON_MAX_K 64 /* k + (sname == 'train') <= 64 in lemon_neighbors / one scan pass; the reference grid tops out at 50+1 (experiments.py:86) */
#define LEMON_MAX_K_DEEP 2048 /* lemon_index_search accepts deeper lists (faiss has no limit): passes of LEMON_MAX_K */

/* search algorithm selector for lemon_index_set_algo() */
#define LEMON_ALGO_AUTO 0
#define LEMON_ALGO_F32_MFMA 1   /* exact fp32 MFMA scan (v_mfma_f32_32x32x2_f32)                       */
#define LEMON_ALGO_BF16_FILTER 2 /* 16-bit MFMA filter with a rigorous error band + exact fp32 re-rank; */
                                 /* returns bit-identical results to LEMON_ALGO_F32_MFMA.  The filter   */
                                 /* operands are fp16 copies since round 3 (bf16 before: name kept)     */
#define LEMON_ALGO_F16_FILTER LEMON_ALGO_BF16_FILTER

typedef struct lemon_index lemon_index_t;

const char *lemon_last_error(void);
/* library / device facts, for logs: returns 0 and fills what it can */
int lemon_version(int *major, int *minor);

/* ---- row-wise helpers ------------------------------------------------------------ */

/* lib/utils/utils.py:39-40 normalize_vectors == F.normalize(p=2, dim=1, eps=1e-12);
 * call sites run_lemon.py:163-164,230-233.  y_dev may alias x_dev. */
int lemon_normalize_rows(const float *x_dev, int64_t n, int d, float *y_dev, void *stream);

/* run_lemon.py:169 (IP: 1 - <a_i,b_i>), :173 (L2: sum_k (a_ik-b_ik)^2) -> dists_tr;
 * run_lemon.py:250-253 -> d_1.  out_dev [n]. */
int lemon_paired_distance(int metric, const float *a_dev, const float *b_dev, int64_t n, int d,
                          float *out_dev, void *stream);

/* DistanceEvaluator.our_metric, lib/metrics/distance_metrics.py:48-73 (used by
 * lib/baselines/run_clip_sim.py:235-248): paired distance of row i of a and row i of b WITHOUT
 * assuming normalised inputs.  kind 0: cosine 1 - <a,b>/(|a||b|); 1: euclidean (NOT squared);
 * 2: manhattan.  The reference takes the diagonal of the full n x n pairwise matrix. */
int lemon_paired_metric(int kind, const float *a_dev, const float *b_dev, int64_t n, int d,
                        float *out_dev, void *stream);

/* --normalize_d1, run_lemon.py:244-248: d1[i] = softmax_c(dist(img_i, cls_txt_c))[noisy_label[i]].
 * cls_txt_dev [C,d] (run_lemon.py:180-190), noisy_label_dev [n] int32. */
int lemon_d1_normalized(int metric, const float *q_img_dev, int64_t n, int d,
                        const float *cls_txt_dev, int C, const int32_t *noisy_label_dev,
                        float *d1_dev, void *stream);

/* Zero-shot "CLIP logits" baseline, lib/baselines/train_zero_shot_clip_baseline.py:207-224: per image,
 * conf[i] = softmax_c(1 - dist(cls_txt_c, img_i))[noisy_label[i]] with dist = DistanceEvaluator.our_metric
 * (lib/metrics/distance_metrics.py:48-73) on UN-normalised embeddings: kind 0 = 1 - cosine similarity,
 * 1 = euclidean (not squared), 2 = manhattan.  img_dev [n,d], cls_txt_dev [C,d], C <= 1024. */
int lemon_class_confidence(int kind, const float *img_dev, int64_t n, int d, const float *cls_txt_dev, int C,
                           const int32_t *noisy_label_dev, float *conf_dev, void *stream);

/* generic_transform (lib/datasets/utils.py:159-170: Resize(224, BICUBIC) -> CenterCrop(224) -> ToTensor ->
 * Normalize) for a batch of equally sized uint8 HWC images, bit-identical to PIL + torch: the two integer
 * resampling passes of PIL (22-bit fixed-point taps, horizontal then vertical, clip8) and (v/255-mean)/std.
 * img_dev [batch, in_h, in_w, 3] uint8; kk_*_dev [out_size, ks_*] int32 taps and bnd_*_dev [out_size, 2]
 * (first input index, tap count) for the out_size CROPPED output columns / rows (built on the host:
 * lemon_amd/data.py::pil_bicubic_tables); rows_per_block output rows per workgroup, max_rows_per_block
 * = the largest number of input rows one block's vertical windows span (x out_size x 3 bytes <= 64 KB);
 * mean3_host/std3_host: 3 floats each (host memory); out_dev [batch, 3, out_size, out_size] float32 when
 * patch == 0, else patch-major [batch, (out_size/patch)^2, 3*patch*patch] -- the rows the ViT patch
 * embedding (a stride == kernel convolution, HF CLIPVisionEmbeddings / chexzero_clip.py:226-238) multiplies,
 * so that it becomes one lemon_linear_f32 GEMM with no im2col pass. */
int lemon_preprocess_u8(const uint8_t *img_dev, int64_t batch, int in_h, int in_w, const int32_t *kk_h_dev,
                        const int32_t *bnd_h_dev, int ks_h, const int32_t *kk_v_dev, const int32_t *bnd_v_dev,
                        int ks_v, int out_size, int max_rows_per_block, int rows_per_block,
                        const float *mean3_host, const float *std3_host, int patch, float *out_dev, void *stream);
/* The same transform with the patch rows written as the tile-major fp16 split operand of lemon_linear_f16x3t (rows = batch *
 * (out_size / patch)^2 padded to 128, k = 3 patch^2; patch % 4 == 0, 3 patch^2 % 16 == 0): the patch embedding then runs in the
 * hand-written GEMM straight from this kernel's output -- no fp32 pixel tensor, no split pass.  Values are exactly the fp16 split
 * (split3.hpp) of what lemon_preprocess_u8 writes. */
int lemon_preprocess_u8_f16x3t(const uint8_t *img_dev, int64_t batch, int in_h, int in_w, const int32_t *kk_h_dev,
                               const int32_t *bnd_h_dev, int ks_h, const int32_t *kk_v_dev, const int32_t *bnd_v_dev,
                               int ks_v, int out_size, int max_rows_per_block, int rows_per_block,
                               const float *mean3_host, const float *std3_host, int patch, uint16_t *outt_dev, void *stream);

/* Multi-head self-attention of the CLIP towers, fused to one pass: the attention inside
 * encode_image / encode_text (lib/models/downstream_models.py:37-41 -> HF CLIPAttention; in-tree twin
 * lib/models/chexzero_clip.py:191-212 with the causal mask of :348-354 for text).
 * qkv_dev [batch, seq_len, 3, heads, head_dim] float32 = the fused q/k/v projection output;
 * out_dev [batch, seq_len, heads*head_dim] = softmax(q k^T / sqrt(head_dim) [+ causal]) v with heads
 * concatenated, ready for the output projection.  head_dim must be 64, seq_len <= 288.
 * Arithmetic (all lemon_attention_* entry points): by default the two products run as SPLIT products on the fp16 matrix cores
 * -- q, k, v and the probabilities are carried as fp16 pairs (hi = f16(v), lo = the remainder: 22 bits + sign, fp32
 * accumulate; error vs float64 at the level of the all-fp32 form).  RANGE: |q|, |k|, |v| must stay below 65 520, beyond that
 * the fp16 parts are inf and the result NaN (loud, not wrong); PRECISION: the short kernels (seq_len <= 64) scale lo by 2^11
 * (full relative precision), the general kernel stores lo unscaled (absolute precision 2^-25: values below 2^-3 keep less than
 * 22 bits, still >= 2^-25 absolute).  lemon_attention_set_f16(0) switches the CALLING THREAD to v_mfma_f32_32x32x2_f32 for both
 * products (no range limit, the fp32 GEMM modes select it; returns the thread's previous setting); $LEMON_ATTN_F16=0 starts
 * every thread there. */
int lemon_attention_f32(const float *qkv_dev, int64_t batch, int seq_len, int heads, int head_dim,
                        int causal, float *out_dev, void *stream);
int lemon_attention_set_f16(int on);
/* The same attention with the result written as the 3-way bf16 split activation operand of lemon_linear_bf16x6 (below):
 * out6_dev [batch*seq_len, 6*heads*64] bf16, 16-byte aligned.  The fp32 result is split at the store, not recomputed. */
int lemon_attention_split3(const float *qkv_dev, int64_t batch, int seq_len, int heads, int head_dim,
                           int causal, uint16_t *out6_dev, void *stream);

/* LayerNorm over the last dimension, float32: y = (x - mean) / sqrt(var + eps) * weight + bias (biased variance, like
 * torch.nn.LayerNorm) -- layer_norm1/2, pre_layrnorm, post_layernorm, final_layer_norm of the towers behind
 * encode_image / encode_text (lib/models/downstream_models.py:37-41; in-tree twin lib/models/chexzero_clip.py:177-183).
 * x_dev, y_dev [rows, width] (y_dev may alias x_dev), width a multiple of 4 and <= 2048, pointers 16-byte aligned. */
int lemon_layernorm_f32(const float *x_dev, const float *weight_dev, const float *bias_dev, float eps, int64_t rows,
                        int width, float *y_dev, void *stream);

/* Token assembly of the towers, one pass each instead of torch's cat / gather + add (+ LayerNorm):
 *   vision (HF CLIPVisionEmbeddings + pre_layrnorm; lib/models/chexzero_clip.py:243-249):
 *     y[b,0] = LN(cls + pos[0]),  y[b,1+p] = LN(patches[b,p] + pos[1+p]);  patches_dev [batch, n_tokens-1, width] is the
 *     patch-embedding GEMM's output, y_dev [batch, n_tokens, width]; ln_weight_dev = ln_bias_dev = NULL: no LayerNorm (timm's
 *     VisionTransformer._pos_embed has none: norm_pre is the identity in BiomedCLIP's vit_base_patch16_224);
 *   text (token + position embedding, chexzero_clip.py:363-365): y[b,t] = tok_emb[ids[b,t]] + pos[t] for t < seq_len;
 *     ids_dev int64 with row pitch ids_pitch >= seq_len (the caller's [batch, context] id matrix, truncated in place). */
int lemon_vision_tokens_ln(const float *patches_dev, const float *cls_dev, const float *pos_dev,
                           const float *ln_weight_dev, const float *ln_bias_dev, float eps, int64_t batch,
                           int n_tokens, int width, float *y_dev, void *stream);
int lemon_text_tokens(const int64_t *ids_dev, int64_t ids_pitch, const float *tok_emb_dev, const float *pos_dev,
                      int64_t batch, int seq_len, int width, int vocab, float *y_dev, void *stream);

/* Linear layer of the CLIP towers with its element-wise tail fused into the GEMM:
 *   y[m,n] = act(alpha x[m,k] W[n,k]^T + bias[n]) (+ residual[m,n])         float32, row-major
 * (nn.Linear inside HF CLIPEncoderLayer / lib/models/chexzero_clip.py:191-212, driven by
 * lib/models/downstream_models.py:30-41).  The GEMM runs on hipBLASLt; act = LEMON_ACT_SILU
 * (u*sigmoid(u)) and the residual add ride in its epilogue.  QuickGELU z*sigmoid(1.702z)
 * (chexzero_clip.py:186-188) is silu(1.702 z)/1.702: call with alpha = 1.702 and a bias scaled by 1.702,
 * and give the consuming GEMM alpha = 1/1.702 (lemon_amd/clip.py does).  bias_dev / residual_dev may
 * be NULL; residual_dev may alias y_dev; activation and residual cannot be combined.
 * Solution choice per (m,n,k,epilogue,residual) key -- reproducible by default: a key supplied by
 * lemon_linear_load_tuned() uses the recorded hipBLASLt solution index (checked once per process, on first
 * use and on the caller's operands, against the library's first-ranked solution: one extra GEMM, one stream
 * synchronisation; dropped if it disagrees); every other key uses the library's first-ranked supported
 * solution (no timing, no allocation, no synchronisation; identical in every process).  Only after
 * lemon_linear_set_tuning(1) (or LEMON_LINEAR_TUNE=1) is an unknown key benchmarked over all solutions
 * (synchronises; bounded by LEMON_LINEAR_TUNE_MS, default 6000): that is the offline tuner's mode
 * (tools/tune_gemms.py), never the inference path's. */
#define LEMON_ACT_NONE 0
#define LEMON_ACT_SILU 1
/* the exact GELU u/2 (1 + erf(u / sqrt 2)) of the towers of open_clip's BiomedCLIP (lib/models/utils.py:72-78: timm
 * vit_base_patch16_224 blocks and the PubMedBERT layers both use nn.GELU).  In the library GEMMs it runs as one in-place pass
 * behind the bias epilogue (hipBLASLt's own GELU epilogue is the tanh approximation); in lemon_linear_f16x3t(_ln) it rides in
 * the operand epilogue like SiLU. */
#define LEMON_ACT_GELU 2
int lemon_linear_f32(const float *x_dev, const float *w_dev, const float *bias_dev, const float *residual_dev,
                     int64_t m, int n, int k, float alpha, int act, float *y_dev, void *stream);

/* The same nn.Linear (same reference call sites, same epilogues, same fp32 result type) with the products on the bf16
 * matrix cores at fp32-equivalent accuracy: every fp32 operand value v is split EXACTLY into three bf16 parts
 * (hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid): 24 significant bits) and the six cross products of order
 * <= 2 are summed by ONE bf16 GEMM with fp32 accumulation over a 6k-long k axis:
 *     x6_dev [m, 6k] rows = [hi | hi | mid | hi | mid | lo]   (lemon_layernorm_split3, or lemon_split3_f32 with weight = 0)
 *     w6_dev [n, 6k] rows = [hi | mid | hi | lo | mid | hi]   (lemon_split3_f32 with weight = 1; once per weight)
 * k6 = 6k.  The dropped products are O(2^-24) of the result; what the GEMM delivers is bounded by its fp32 accumulation, like
 * the fp32 GEMM's own result (max error vs float64, relative to the largest output, 1.3-2.1e-6 at the ViT-B/32 tower shapes;
 * fp32 GEMM 1.0-2.1e-6 -- tools/split_gemm_probe.py).  y_dev / bias_dev / residual_dev are float32. */
int lemon_linear_bf16x6(const uint16_t *x6_dev, const uint16_t *w6_dev, const float *bias_dev, const float *residual_dev,
                        int64_t m, int n, int k6, float alpha, int act, float *y_dev, void *stream);
/* 3-way bf16 split of a row-major float32 matrix [rows, k] (k a multiple of 4) into lemon_linear_bf16x6's operand rows
 * y6_dev [rows, 6k] bf16; weight = 0: activation layout, 1: weight layout. */
int lemon_split3_f32(const float *x_dev, int64_t rows, int k, int weight, uint16_t *y6_dev, void *stream);
/* lemon_layernorm_f32 whose result is written as the split activation operand y6_dev [rows, 6 width] bf16 (one pass). */
int lemon_layernorm_split3(const float *x_dev, const float *weight_dev, const float *bias_dev, float eps, int64_t rows,
                           int width, uint16_t *y6_dev, void *stream);
/* The same nn.Linear once more, with HALF the matrix work of lemon_linear_bf16x6: every fp32 operand value v is split into
 * two fp16 parts, hi = f16(v) and lo = v - hi (exact in fp32; kept as f16(lo * 2^11): 11 + 11 significant bits and lo's sign,
 * |v - hi - lo| <= 2^-23 |v|), and ONE fp16 GEMM with fp32 accumulation over a 3k-long k axis sums hi.hi + hi.lo + lo.hi:
 *     x3_dev [m, 3k] rows = [hi | hi | lo 2^11]            (lemon_layernorm_f16x3 / lemon_attention_f16x3 / lemon_split_f16x3)
 *     w3_dev [n, 3k] rows = [hi | lo | hi 2^-11] of w * wscale   (lemon_split_f16x3 with weight = 1; once per weight)
 * wscale is a power of two chosen by the caller so that max |w| * wscale lies in [2^14, 2^15) (lo and hi 2^-11 of every
 * weight that matters are then fp16 normals); the caller passes alpha / wscale as `alpha`.  k3 = 3k.  The dropped lo.lo
 * product is <= 2^-22 (typically 2^-26) of a product: against float64 the result is as accurate as the fp32 GEMM's
 * (tools/split_gemm_probe.py).  Operand values with |v| >= 65 520 (beyond fp16) turn the outputs they reach into NaN. */
int lemon_linear_f16x3(const uint16_t *x3_dev, const uint16_t *w3_dev, const float *bias_dev, const float *residual_dev,
                       int64_t m, int n, int k3, float alpha, int act, float *y_dev, void *stream);
/* 2-way fp16 split of a row-major float32 matrix [rows, k] (k a multiple of 4) into lemon_linear_f16x3's operand rows
 * y3_dev [rows, 3k] fp16; weight = 0: activation layout (wscale ignored), 1: weight layout of x * wscale (a power of two). */
int lemon_split_f16x3(const float *x_dev, int64_t rows, int k, int weight, float wscale, uint16_t *y3_dev, void *stream);
/* lemon_layernorm_f32 / lemon_attention_f32 whose result is written as the fp16 split activation operand (one pass):
 * y3_dev [rows, 3 width], out3_dev [batch*seq_len, 3*heads*64], 16-byte aligned. */
int lemon_layernorm_f16x3(const float *x_dev, const float *weight_dev, const float *bias_dev, float eps, int64_t rows,
                          int width, uint16_t *y3_dev, void *stream);
int lemon_attention_f16x3(const float *qkv_dev, int64_t batch, int seq_len, int heads, int head_dim,
                          int causal, uint16_t *out3_dev, void *stream);
/* The MLP of a transformer block (fc1 -> QuickGELU -> fc2; lib/models/downstream_models.py:37-41 via HF CLIPMLP, in-tree twin
 * lib/models/chexzero_clip.py:171-183) with the arithmetic of lemon_linear_f16x3 in a HAND-WRITTEN gfx950 kernel whose operands
 * are tile-major in MFMA fragment order:
 *     activation operand at_dev: ceil(m / 128) * 128 rows x k, halves [row tile of 128][k / 16][hi, lo 2^11][row block of 32]
 *                                [k half][row in block][8 k]  (4 bytes per element; rows >= m need not be initialised)
 *     weight operand wt_dev:     n rows x k of w * wscale, the same with 256-row tiles and parts hi, lo   (lemon_pack_weight_f16x3t)
 * lemon_layernorm_f16x3t writes the activation operand (the LayerNorm in front of fc1); lemon_linear_f16x3t computes
 *     act = LEMON_ACT_NONE, out_operand = 0:  out_dev fp32 [m, n] = alpha * x W^T + bias (+ residual)          (fc2)
 *     act = LEMON_ACT_SILU, out_operand = 1:  out_dev = the activation operand (k' = n) of silu(alpha * x W^T + bias)   (fc1:
 *         the [m, n] fp32 tensor and the split pass over it never exist)
 *     act = LEMON_ACT_GELU, out_operand = 1:  the same with the exact GELU (BiomedCLIP's towers)
 * with n a multiple of 256 and k a multiple of 16; the caller folds 1 / wscale into alpha.  Results are independent of a row's
 * position in the batch (fixed k order, no split-k).  lemon_unpack_act_f16x3t turns an activation operand back into fp32
 * (hi + lo 2^-11; tests). */
int lemon_pack_weight_f16x3t(const float *w_dev, int n, int k, float wscale, uint16_t *wt_dev, void *stream);
int lemon_layernorm_f16x3t(const float *x_dev, const float *weight_dev, const float *bias_dev, float eps, int64_t rows,
                           int width, uint16_t *yt_dev, void *stream);
int lemon_linear_f16x3t(const uint16_t *at_dev, const uint16_t *wt_dev, const float *bias_dev, const float *residual_dev,
                        int64_t m, int n, int k, float alpha, int act, int out_operand, void *out_dev, void *stream);
int lemon_unpack_act_f16x3t(const uint16_t *at_dev, int64_t rows, int k, float *y_dev, void *stream);
/* LayerNorm folded into the hand-written GEMMs (HF CLIPEncoderLayer.layer_norm1 / layer_norm2 in front of q/k/v_proj and
 * mlp.fc1; in-tree twin lib/models/chexzero_clip.py:171-183: ln_1, ln_2).  LN(x) W^T + b = rstd (x W'^T - mean c) + b' with
 * W' = W diag(gamma), c = W' 1, b' = b + W beta, so the GEMM behind a LayerNorm takes the RAW residual stream as its operand:
 *   lemon_linear_f16x3t_ln(..., row_aff_dev, colsum_dev, NULL, NULL):  out = act(row_aff[m].x * (alpha x W'^T) + row_aff[m].y *
 *       colsum[n] + bias[n]) (+ residual) with row_aff [m, 2] = (rstd, -mean rstd) and colsum [n] = alpha * sum_k W'[n, k]
 *       (of the PACKED weight: hi + lo), bias = b'
 *   lemon_linear_f16x3t_ln(..., NULL, NULL, emit_t_dev, emit_stats_dev) (act none, fp32 out):  the fp32 result is ALSO written as
 *       the tile-major activation operand emit_t_dev (ceil(m / 128) * 128 rows x n) and emit_stats_dev [m, n / 128, 2] receives
 *       per row and 128-column group (mean, sum of squared deviations); lemon_ln_finalize merges them into row_aff
 *   lemon_rowstats_f16x3t: operand + row_aff of a tensor no GEMM produced (a tower's first block)
 * k must be a multiple of 32.  With all four extra pointers NULL the call is lemon_linear_f16x3t.
 * Accuracy contract: the fold's rounding error is the un-folded GEMM's times sqrt(1 + mean^2 / var) of the row, so rows with
 * |mean| rstd > 8 (LEMON_LN_FOLD_MAX_SHIFT) get row_aff = (NaN, NaN) from lemon_ln_finalize / lemon_rowstats_f16x3t: their
 * output rows come out non-finite and the caller must redo them without the fold (lemon_amd.pipeline.Embedder re-embeds the
 * micro-batch with LayerNorm kernels). */
int lemon_linear_f16x3t_ln(const uint16_t *at_dev, const uint16_t *wt_dev, const float *bias_dev, const float *residual_dev,
                           int64_t m, int n, int k, float alpha, int act, int out_operand, void *out_dev, const float *row_aff_dev,
                           const float *colsum_dev, uint16_t *emit_t_dev, float *emit_stats_dev, void *stream);
/* The producing GEMM of a block chain (HF CLIPEncoderLayer: self_attn.out_proj and mlp.fc2 with their residual adds;
 * lib/models/chexzero_clip.py:191-212) in its leanest form -- lemon_linear_f16x3t_ln's emit side with two options:
 *   residual_t_dev  the residual as the tile-major activation operand [m, n] an emitting GEMM in front left (x = hi + lo 2^-11:
 *                   22 significant bits -- one more rounding of the size the split products make anyway) instead of residual_dev
 *                   (fp32 [m, n]); at most one of the two may be given
 *   out_dev = NULL  no fp32 result: every consumer reads emit_t_dev (the next GEMM as its operand, the one behind it as its
 *                   residual) -- the output projection then writes 6 instead of 10 bytes per element
 * emit_t_dev / emit_stats_dev as in lemon_linear_f16x3t_ln (both required); act none; k a multiple of 32. */
int lemon_linear_f16x3t_chain(const uint16_t *at_dev, const uint16_t *wt_dev, const float *bias_dev, const float *residual_dev,
                              const uint16_t *residual_t_dev, int64_t m, int n, int k, float alpha, float *out_dev,
                              uint16_t *emit_t_dev, float *emit_stats_dev, void *stream);
int lemon_ln_finalize(const float *partials_dev, int64_t rows, int width, float eps, float *row_aff_dev, void *stream);
int lemon_rowstats_f16x3t(const float *x_dev, float eps, int64_t rows, int width, uint16_t *yt_dev, float *row_aff_dev, void *stream);
/* Timing of the hand-written GEMM (no reference counterpart; bench.py's roofline object): while profiling is on every
 * lemon_linear_f16x3t launch of this process is bracketed by HIP events on its launch stream (a pool that grows with the
 * launches between two reads; a launch that cannot get its events fails with LEMON_E_HIP, none is silently left out).
 * profile_read waits for the recorded events, returns the number
 * of bracketed launches, their summed durations and their summed arithmetic (2 m n 3k: the three fp16 products the kernel
 * executes per fp32 product), and rewinds the pool. */
int lemon_linear_f16x3t_set_profiling(int on);
/* The matrix instruction of lemon_linear_f16x3t: 16 = v_mfma_f32_16x16x32_f16 (default wherever k is a multiple of 32), 32 =
 * v_mfma_f32_32x32x16_f16 (always used for the other k), 0 = back to the default ($LEMON_GEMM_MFMA).  Same arithmetic, another
 * summation order inside a k32 step; process-wide, for A/B runs and tests.  Returns the previous setting (>= 0) or an error. */
int lemon_linear_f16x3t_set_mfma(int shape);
int lemon_linear_f16x3t_profile_read(int64_t *launches, double *kernel_ms, double *flops);
/* lemon_attention_f32 whose result is written as that activation operand (rows = batch*seq_len, k = heads*64): the output
 * projection then runs in the hand-written GEMM too (with QKV: all four GEMMs of a block). */
int lemon_attention_f16x3t(const float *qkv_dev, int64_t batch, int seq_len, int heads, int head_dim,
                           int causal, uint16_t *outt_dev, void *stream);
/* Recorded solution choices: a "# lemon_linear hipblaslt=<version> arch=<gfx name>" stamp line followed by
 * "m,n,k,epilogue,residual,index,usec" lines.  load returns the number of keys taken -- 0 when the stamp
 * does not match this process's hipBLASLt version / device arch (the file is then ignored) -- and dump the
 * number written (<0 on error).  lemon_linear_stamp reports the stamp of the current device. */
int lemon_linear_load_tuned(const char *path);
int lemon_linear_dump_tuned(const char *path);
int lemon_linear_set_tuning(int enabled);
int lemon_linear_stamp(int *hipblaslt_version, char *arch, int arch_len);

/* ---- flat index (faiss.IndexFlatIP / IndexFlatL2 as used by run_lemon.py) ---------- */

/* faiss.IndexFlatIP(d) / faiss.IndexFlatL2(d): run_lemon.py:167-168,171-172;
 * lib/baselines/discrepancy_baseline.py:150-155.  Binds to the current HIP device. */
int lemon_index_create(int metric, int d, lemon_index_t **out);
int lemon_index_free(lemon_index_t *idx);

/* index.add(x): run_lemon.py:175-176.  Copies x (like faiss) and builds the kernel-side
 * layouts; may be called repeatedly (appends).  Allocates device memory: NOT capturable
 * in a hipGraph, and it synchronises `stream` when it has to grow its storage. */
int lemon_index_add(lemon_index_t *idx, const float *x_dev, int64_t n, void *stream);

int64_t lemon_index_ntotal(const lemon_index_t *idx); /* faiss index.ntotal */
int lemon_index_dim(const lemon_index_t *idx);        /* faiss index.d      */
/* device pointer to the stored row-major copy [ntotal, d] (valid until the next add/free) */
const float *lemon_index_data(const lemon_index_t *idx);

/* index.search(x, k): run_lemon.py:235-236; lib/baselines/discrepancy_baseline.py:166,209.
 * D_dev [nq,k] float32 and I_dev [nq,k] int64, best first (IP: descending inner product;
 * L2: ascending SQUARED distance max(0, |q|^2+|x|^2-2<q,x>)).  Slots beyond ntotal get
 * I=-1 and D=-FLT_MAX (IP) / +FLT_MAX (L2).  1 <= k <= LEMON_MAX_K_DEEP (2048; faiss itself has no limit): up to
 * LEMON_MAX_K (64) one scan pass; deeper lists are produced LEMON_MAX_K at a time by key-bounded passes of the exact
 * fp32 scan (each pass admits only rows ranked behind the previous pass's last result: same order, same tie rule).
 * Grows an internal workspace on first use of a larger (nq,k): that first call is not
 * graph-capturable; later calls with nq,k no larger only enqueue kernels.  With LEMON_ALGO_AUTO
 * (default) the first LARGE search (ntotal >= 65536, nq*ntotal >= 8e9, d <= 768) after an add()
 * runs a small probe search and synchronises the stream once to choose between the two scans. */
int lemon_index_search(lemon_index_t *idx, const float *q_dev, int64_t nq, int k,
                       float *D_dev, int64_t *I_dev, void *stream);
int lemon_index_set_algo(lemon_index_t *idx, int algo);
/* Query de-duplication (on by default; LEMON_QUERY_DEDUP=0 disables it process-wide): from 1024 queries on, the rows of
 * q are grouped by content (64-bit hash + stable sort + full bitwise comparison) and, when at most half of them are
 * distinct, the search runs once per distinct row and its (D, I) lists are copied to every member -- exact, because a
 * query's result does not depend on the other queries.  Classification datasets make the text-side search of
 * run_lemon.py:236 a C-query problem this way.  Costs one stream synchronisation per search call of >= 1024
 * queries (the group count is read back to size the reduced search) -- the one exception to "later calls only enqueue
 * kernels" above; lemon_index_set_query_dedup(idx, 0) restores it.  lemon_neighbors applies de-duplication to its TEXT
 * index only (image queries of the LEMoN loop are distinct), so its image-side search never synchronises. */
int lemon_index_set_query_dedup(lemon_index_t *idx, int enabled);

/* last search's dominant-kernel launch statistics (for bench/roofline bookkeeping) */
typedef struct {
    int algo;            /* LEMON_ALGO_* actually used                       */
    int grid, block;     /* launch geometry of the scan kernel               */
    int query_panel;     /* query rows per workgroup (B of the scan model)   */
    int db_splits;       /* how many workgroups share one query panel        */
    int64_t nq, n;
    int d, k;
    int64_t nq_distinct; /* query rows actually searched (== the call's nq unless duplicates were folded) */
} lemon_search_info_t;
int lemon_index_last_search_info(const lemon_index_t *idx, lemon_search_info_t *out);

/* Optional in-library timing of the scan kernel (the dominant kernel of search): when enabled,
 * every scan launch is bracketed by hipEvents recorded on the launch stream.
 * lemon_index_profile_read synchronises those events (host sync!) and returns, summed since the
 * last reset: launches, kernel milliseconds, algorithmic flops (2*nq*n*d) and algorithmic HBM bytes
 * of the scan model (4*d*(nq + ceil(nq/B)*n) + 12*k*nq with B = query_panel); then
 * resets the counters.  Any output pointer may be NULL. */
int lemon_index_set_profiling(lemon_index_t *idx, int enabled);
int lemon_index_profile_read(lemon_index_t *idx, int64_t *launches, double *kernel_ms,
                             double *algo_flops, double *algo_bytes);

/* Host-only view of how the exact scan (LEMON_ALGO_F32_MFMA) decomposes `panels` query panels (128 queries each) x
 * `n_tiles` database tiles (128 rows each) over its workgroups: for tests and tools, no device is touched.
 *   seg_begin [*grid + 1]  first segment of every workgroup (cap_wgs + 1 ints of room)
 *   pieces    [panels]     pieces the panel is scanned in (what the merge combines)
 *   segs      [4 * *n_segs] (panel, first tile, tiles, piece number inside the panel) per segment (cap_segs segments of room)
 * Returns LEMON_E_ARG when an output does not fit. */
int lemon_debug_scan_plan(int panels, int n_tiles, int *grid, int *splits, int *seg_begin, int cap_wgs,
                          int *pieces, int *segs, int cap_segs, int *n_segs);

/* ---- multimodal neighbours (the per-sample loop run_lemon.py:238-307) ------------- */

/*
 * One call = one split of the reference's scoring loop.
 *   idx_img / idx_txt : indices over the normalised DB image / text embeddings (same
 *                       ntotal, d and metric); dists_tr_dev [ntotal] (run_lemon.py:169/173)
 *   q_img_dev, q_txt_dev [nq,d] normalised query embeddings (run_lemon.py:230-233)
 *   drop_self : 1 on the train split -- search k+1 (:235-236) then drop result[0] where
 *               in_db_dev[i] != 0 else result[-1] (:257-263,277-283); in_db_dev [nq] u8,
 *               may be NULL (= all ones) and is ignored when drop_self == 0
 *   discrete  : --use_discrete_for_text (:266-267): dists_n = 1 - [tr_label_id[I_n] == q_label_id]
 *               (int32 ids of the prompt strings; may be NULL when discrete == 0)
 * Outputs, caller-allocated: d1 [nq]; D_n, dists_n, dists_tr_n, D_m, dists_m, dists_tr_m [nq,k]
 * float32 with the sign convention of :269-270,285-286; I_n, I_m [nq,k] int64 (may be NULL).
 * Neighbours that do not exist (ntotal < k+drop_self) give I=-1 and NaN distances.
 * Limit: k + drop_self <= LEMON_MAX_K (64), i.e. k <= 63 on the train split -- the faiss contract behind :235-236 has no
 * such limit, but the loop's per-sample record is built from ONE scan pass here; the reference's own grids stop at
 * k = 50 (+1) (experiments.py:86), and deeper lists are available from lemon_index_search (k <= 2048).  Larger k is
 * refused with LEMON_E_INVALID, never truncated.  De-duplication of identical query rows is applied to idx_txt only.
 */
int lemon_neighbors(lemon_index_t *idx_img, lemon_index_t *idx_txt, const float *dists_tr_dev,
                    const float *q_img_dev, const float *q_txt_dev, int64_t nq, int k,
                    int drop_self, const uint8_t *in_db_dev, int discrete,
                    const int32_t *tr_label_id_dev, const int32_t *q_label_id_dev,
                    float *d1_dev, float *D_n_dev, float *dists_n_dev, float *dists_tr_n_dev,
                    int64_t *I_n_dev, float *D_m_dev, float *dists_m_dev, float *dists_tr_m_dev,
                    int64_t *I_m_dev, void *stream);

/* Discrepancy baselines, lib/baselines/discrepancy_baseline.py:164-242 (next-row scope).
 * method 0 = dis_x / dis_y: mean cosine distance between the query's embedding qv_dev [nq,d] and the
 * second-order neighbours (through the DB's own text kNN cache of k+1, self removed by index, :165-169)
 * of its k (+1 on train, not dropped) text neighbours; method 1 = div_x / div_y: sum of pairwise cosine
 * distances inside the neighbour set divided by k^2.  E_tr_dev [ntotal,d] are the DB embeddings of the
 * scored modality (x: image, y: text); neighbours always come from idx_txt (:209).  out_dev [nq]. */
int lemon_discrepancy(int method, lemon_index_t *idx_txt, const float *E_tr_dev, const float *qv_dev,
                      const float *q_txt_dev, int64_t nq, int k, int is_train, float *out_dev, void *stream);

/* lib/metrics/utils.py:47-82 calc_scores_given_hparams_vectorized (== loop twin :21-45):
 * score = d_1 + beta*mean_j[e^{-tau_1_n D_n} e^{-tau_2_n dists_tr_n} dists_n] + gamma*(same for m).
 * hp = {beta, gamma, tau_1_n, tau_2_n, tau_1_m, tau_2_m} (host doubles, passed by value in the
 * launch).  score_dev [n] float64; d_n_dev / d_m_dev [n] float64 or NULL. */
int lemon_score(const float *d1_dev, const float *D_n_dev, const float *dists_tr_n_dev,
                const float *dists_n_dev, const float *D_m_dev, const float *dists_tr_m_dev,
                const float *dists_m_dev, int64_t n, int k, const double hp[6],
                double *score_dev, double *d_n_dev, double *d_m_dev, void *stream);

/* The hyper-parameter grid of run_lemon.py:319-384 as one batch: for each of G rows of hp_dev
 * [G,6] = (beta, gamma, tau_1_n, tau_2_n, tau_1_m, tau_2_m) the scores of lemon_score (same float64
 * arithmetic) and optimize_f1_efficient (lib/metrics/utils.py:286-296: scipy fminbound on -F1(y, score >= t),
 * xtol, maxfun) -> f1_dev[G], thres_dev[G] (float64), bit-identical to evaluating the grid points one by one
 * on the host.  y_dev [n] uint8 (is_mislabel); scores_ws_dev: caller-provided [G, n] float64 workspace;
 * G <= 65535.  A grid point whose scores are not all finite gets f1 = 0, thres = NaN. */
int lemon_grid_f1(const float *d1_dev, const float *D_n_dev, const float *dists_tr_n_dev, const float *dists_n_dev,
                  const float *D_m_dev, const float *dists_tr_m_dev, const float *dists_m_dev,
                  const uint8_t *y_dev, int64_t n, int k, const double *hp_dev, int G, double xtol, int maxfun,
                  double *scores_ws_dev, double *f1_dev, double *thres_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LEMON_HIP_H */
