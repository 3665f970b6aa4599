/*
 * t2s_hip.h -- C ABI of libt2s_hip.so, the MI355X (gfx950) kernels of the T2S-QA fusion path.
 *
 * The reference (zhousheng97/ViTXT-GQA) is pure PyTorch and has NO native ABI for this path
 * (SURVEY.md section 2.1); the boundary below is therefore defined by this build, one entry
 * point per op group the reference executes on the hot path.  Each declaration cites the
 * reference code it replaces (paths relative to the reference checkout).
 *
 * Conventions:
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller;
 *   - return 0 on success, non-zero on error; t2s_last_error() gives the thread-local message;
 *   - no allocation, no retained pointers, no global mutable state, no hipSetDevice, no
 *     implicit synchronisation: every kernel is enqueued asynchronously on `stream`
 *     (a hipStream_t passed as void*); the caller supplies outputs and workspaces;
 *   - dtype: T2S_F32 = 0, T2S_BF16 = 1 (activation/storage type; all reductions, softmax and
 *     LayerNorm statistics are fp32 in both modes);
 *   - "rows" are token rows of 768 = 12 heads x 64; strides are in ELEMENTS.
 */
#ifndef T2S_HIP_H
#define T2S_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2S_ABI_VERSION 6
#define T2S_F32 0
#define T2S_BF16 1
#define T2S_HEAD_DIM 64

typedef void* t2s_stream_t; /* hipStream_t */

int t2s_abi_version(void);
const char* t2s_last_error(void);

/* ---- key compaction ------------------------------------------------------------------
 * Replaces the materialised [B,1,L,L] additive mask of QTV.forward / MMT.forward
 * (pythia/models/t2s.py:413-419, 609-618): every query row of a sample sees the same key
 * set, so the mask is a per-sample list of visible key rows.  valid: [B, L] uint8 (0/1) over
 * the prefix rows; out_idx: [B, idx_cap] int32 gets the ascending row numbers of the valid
 * keys followed by the n_dec decoder rows dec_row0 .. dec_row0+n_dec-1; out_cnt: [B] int32 =
 * number of valid prefix keys.  idx_cap >= L + n_dec.  Skipping masked keys is exact in fp32:
 * exp(-10000 + s - max) == 0. */
int t2s_compact_keys(const uint8_t* valid, int32_t* out_idx, int32_t* out_cnt, int B, int L,
                     int idx_cap, int n_dec, int dec_row0, t2s_stream_t stream);

/* ---- BERT self-attention (third-party pytorch_transformers BertSelfAttention; call sites
 * t2s.py:423-427, 538-542, 622-626):  softmax(Q K^T * scale + M) V, M = 0/-10000 key mask,
 * plus the 12x12 causal tail of MMT.  Flash-style (no L x L buffer), MFMA on gfx950.
 *
 * q/k/v/out: element (b, row, head, d) at  base + b*batch_stride + row*row_stride + head*64 + d
 * (so a fused [B, L, 3*768] QKV buffer is addressed with row_stride = 2304).
 * kv_idx [B, idx_cap] / kv_cnt [B] as produced by t2s_compact_keys; kv_idx == NULL means the
 * dense key list 0..idx_cap-1 whose last n_dec rows are decoder keys.
 * Decoder rule: key position p >= kv_cnt[b] is decoder step j = p - kv_cnt[b]; it is visible to
 * query row r iff r - dec_q0 >= j.
 * lse: [B, H, Lq] fp32, natural-log-sum-exp of the scaled scores (saved for backward).
 * drop_p > 0: attention-probability dropout (BertSelfAttention: dropout(softmax(.)) before .V):
 * out = (softmax(.) * keep / (1 - p')) V, keep(b, h, q, list position) a stateless function of drop_seed
 * (vitxt_gqa_amd/csrc/attn_common.h), p' = round(65536 p)/65536 (a 16-bit threshold: 0.1 -> 0.100006); the mask needs no workspace.  The backward call
 * regenerates the mask from the same seed; t2s_attn_dropout_mask exports it. */
int t2s_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
                 const int32_t* kv_idx, const int32_t* kv_cnt,
                 int B, int H, int Lq, int idx_cap, int n_dec, int dec_q0,
                 int64_t q_row_stride, int64_t q_batch_stride,
                 int64_t kv_row_stride, int64_t kv_batch_stride,
                 int64_t o_row_stride, int64_t o_batch_stride,
                 float scale, int dtype, float drop_p, uint64_t drop_seed, t2s_stream_t stream);
/* keep mask [B, H, Lq, Lk] (0/1 bytes; Lk = key-list positions) of the dropout above (tests). */
int t2s_attn_dropout_mask(uint8_t* out, int B, int H, int Lq, int Lk, float drop_p, uint64_t drop_seed,
                          t2s_stream_t stream);

/* Backward of the above.  delta: [B, H, Lq] fp32 workspace (rowsum(dO * O), written here).
 * dq/dk/dv use the q/kv strides.  dk/dv rows of keys that are not in the key list are NOT
 * written: the caller zero-fills dk/dv (their gradient is exactly 0 in the reference too).
 * max_keys: host-known upper bound on kv_cnt[b] + n_dec (<= idx_cap; pass idx_cap if unknown);
 * it only sizes the dK/dV grid (key blocks past a sample's list exit immediately either way). */
int t2s_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                 const float* lse, float* delta, void* dq, void* dk, void* dv,
                 const int32_t* kv_idx, const int32_t* kv_cnt,
                 int B, int H, int Lq, int idx_cap, int n_dec, int dec_q0, int max_keys,
                 int64_t q_row_stride, int64_t q_batch_stride,
                 int64_t kv_row_stride, int64_t kv_batch_stride,
                 int64_t o_row_stride, int64_t o_batch_stride,
                 float scale, int dtype, float drop_p, uint64_t drop_seed, t2s_stream_t stream);
/* The same for the self-attention layout (query rows = the idx_cap - n_dec prefix rows, then decoder rows, of which
 * rows [dec_q0, dec_q0 + n_dec) are this call's; Lq >= idx_cap - in the shared-prefix layout of the three MMT passes
 * of t2s.py:293-313 the decoder rows of the other two passes follow the prefix too and are never keys here; bf16),
 * with the zero-fill done here: row_valid [B, idx_cap - n_dec] bytes is the mask the key list was compacted from
 * (t2s_compact_keys' input); the dQ kernel, which visits every (row, head) anyway, writes the zero dK / dV
 * slices of the rows no list entry points at, so dq/dk/dv may be uninitialised memory on entry. */
int t2s_attn_bwd_fill(const void* q, const void* k, const void* v, const void* out, const void* dout,
                      const float* lse, float* delta, void* dq, void* dk, void* dv,
                      const int32_t* kv_idx, const int32_t* kv_cnt, const uint8_t* row_valid,
                      int B, int H, int Lq, int idx_cap, int n_dec, int dec_q0, int max_keys,
                      int64_t q_row_stride, int64_t q_batch_stride,
                      int64_t kv_row_stride, int64_t kv_batch_stride,
                      int64_t o_row_stride, int64_t o_batch_stride,
                      float scale, int dtype, float drop_p, uint64_t drop_seed, t2s_stream_t stream);
/* The same gradients from ONE key-stationary kernel that runs the algorithm's five matrix products per (query, key) pair
 * (S and dP are computed once; the two-kernel form above recomputes them: seven): bf16; attention dropout as above (same mask).
 * dQ is a sum over the 384-key blocks of a (sample, head); dq_mode picks how it is formed:
 *   1  ordered hand-off (default of the Python layer): the key blocks of a pair add their tiles in block order, each reading
 *      the running fp32 sum of its predecessors and storing the new one with plain stores, the last one writing bf16 dq
 *      itself - no zero fill, no cast pass, dq BIT-REPRODUCIBLE.  The key blocks of a pair run on ONE XCD (workgroups of equal
 *      blockIdx % 8), and the sums stay in that XCD's L2 (write-back stores, sc1 loads).  The kernel checks the premise - it ORs
 *      every workgroup's XCC_ID into a word per group - and sets status bit 1 when a group ran on two XCDs;
 *      | 0x200: write-through (sc1) stores instead, correct under any workgroup placement (no check; 2.8 % slower, 2.8 x the
 *      HBM traffic): the form for a device that does not place workgroups round-robin over its XCDs;
 *   0  fp32 atomics into a [B, Lq, H*64] buffer + a cast pass (rounds 2-3): dq depends on arrival order in its last bits.
 * dk / dv are deterministic either way.  workspace: t2s_attn_bwd_fused_workspace_bytes(B, H, Lq) bytes of device memory, contents
 * on entry irrelevant (the call clears what it needs); word 24 of it (uint32) is a status word: bit 0 set = a bounded spin of
 * the hand-off timed out (cannot happen unless a workgroup died).  The launch still ends, and the dq rows of that (sample, head)
 * pair are NaN from the timed-out block on - never a silently wrong number; bit 1 set = the XCD-local sums' placement premise
 * was violated (dq may hold stale sums: discard the step); t2s_status_accumulate / t2s_status_gate below carry the word to
 * the optimizer step.  Tests only: dq_mode 0x101 = the hand-off with a dead predecessor - no block publishes, every successor's
 * wait times out at once; 0x401 = the placement check is fed two XCD numbers per group.
 * row_valid as for t2s_attn_bwd_fill, or NULL (then the caller zero-fills dk / dv).  A sample whose list is empty (kv_cnt = 0 and
 * n_dec = 0) gets exactly zero dq rows in either form. */
int64_t t2s_attn_bwd_fused_workspace_bytes(int B, int H, int Lq);
int t2s_attn_bwd_fused(const void* q, const void* k, const void* v, const void* out, const void* dout,
                       const float* lse, float* delta, void* dq, void* dk, void* dv,
                       void* workspace, int64_t workspace_bytes, int dq_mode,
                       const int32_t* kv_idx, const int32_t* kv_cnt, const uint8_t* row_valid,
                       int B, int H, int Lq, int idx_cap, int n_dec, int dec_q0, int max_keys,
                       int64_t q_row_stride, int64_t q_batch_stride,
                       int64_t kv_row_stride, int64_t kv_batch_stride,
                       int64_t o_row_stride, int64_t o_batch_stride,
                       float scale, int dtype, float drop_p, uint64_t drop_seed, t2s_stream_t stream);

/* ---- residual + LayerNorm (BertSelfOutput / BertOutput / BertLayerNorm; also
 * t2s.py:87-88,116-117,685-687): z = x + res (res may be NULL); y = (z-mean)/sqrt(var+eps)*g+b,
 * biased variance, eps inside the sqrt.  rows x 768.
 * x has x_dtype (a GEMM output); res, y, z_out have stream_dtype (the residual stream: fp32 in
 * both compute modes, so bf16 rounding never touches the running hidden state).  y_lo (may be
 * NULL) receives a bf16 copy of y for the next GEMM; y may be NULL when only y_lo is wanted.  z_out (may be NULL; may alias x when the
 * two dtypes are equal) keeps the pre-norm sum for backward; stats: [rows, 2] fp32 (mean, rstd),
 * may be NULL.  Supported (x, stream): (f32,f32), (bf16,f32), (bf16,bf16).
 * drop_p > 0 applies the hidden-state dropout of BertSelfOutput / BertOutput to x BEFORE the residual
 * add: x <- x * keep / (1 - drop_p), keep(row, col) a stateless hash of (drop_seed, row*768 + col)
 * (t2s_dropout_mask exports it), so the backward call regenerates the mask from the same seed. */
int t2s_add_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                          void* y, void* y_lo, void* z_out, float* stats, int64_t rows, float eps,
                          int x_dtype, int stream_dtype, float drop_p, uint64_t drop_seed,
                          t2s_stream_t stream);

/* The same with the residual given in NORMALISED form: res_z is the pre-LayerNorm sum z of the PREVIOUS
 * residual+LayerNorm block (what that call wrote to z_out), res_stats its (mean, rstd) rows and res_gamma /
 * res_beta its affine; the kernel adds LN(res_z) computed on the fly.  Inside a BERT layer stack the fp32 stream value
 * y is then never written to or re-read from HBM between blocks (y may be NULL when y_lo is given): 12 instead of 16
 * bytes per element of the forward pass.  Replaces the same reference lines as t2s_add_layernorm_fwd. */
int t2s_add_layernorm_fwd_nres(const void* x, const void* res_z, const float* res_stats,
                               const float* res_gamma, const float* res_beta, const float* gamma,
                               const float* beta, void* y, void* y_lo, void* z_out, float* stats,
                               int64_t rows, float eps, int x_dtype, int stream_dtype, float drop_p,
                               uint64_t drop_seed, t2s_stream_t stream);

/* Pre-LayerNorm residual block of the ViT frame-feature producer (tools/video_feat/obtain_vit_feat.py:37-53 runs transformers'
 * ViTLayer: x = x + attn(LN(x)); x = x + mlp(LN(x)); final LayerNorm on the CLS row): in one pass over the rows of the fp32 stream
 * h [rows, width] (row stride h_row_stride elements), h <- h + branch + bias (branch: [rows, width] GEMM output of branch_dtype,
 * bias fp32 [width]; both may be NULL: plain LayerNorm, h untouched), then y = LN(h) * gamma + beta with biased variance and eps
 * inside the sqrt, written as y_dtype (bf16: the next GEMM's operand; f32: the final LayerNorm).  width: a multiple of 4, at most
 * 1280 (ViT-L: 1024). */
int t2s_wide_add_layernorm_fwd(float* h, int64_t h_row_stride, const void* branch, int branch_dtype,
                               const float* bias, const float* gamma, const float* beta, void* y,
                               int y_dtype, int64_t rows, int width, float eps, t2s_stream_t stream);

/* dz = LN backward wrt z (= grad of both x and res); dgamma_part/dbeta_part: [n_part, 768] fp32
 * partial sums (n_part = t2s_layernorm_bwd_parts(rows)); the caller reduces over dim 0.
 * Supported (dy, z, dz) dtypes: (f32,f32,f32), (f32,f32,bf16), (bf16,f32,bf16), (bf16,bf16,bf16).
 * With drop_p > 0, dzx (same dtype as dz, required) receives the gradient of the dropped branch input
 * dz * keep / (1 - drop_p); dz itself is the gradient of the residual input. */
int t2s_layernorm_bwd_parts(int64_t rows);
int t2s_add_layernorm_bwd(const void* dy, const void* z, const float* stats, const float* gamma,
                          void* dz, void* dzx, float* dgamma_part, float* dbeta_part, int64_t rows,
                          int dy_dtype, int z_dtype, int dz_dtype, float drop_p, uint64_t drop_seed,
                          t2s_stream_t stream);
/* The same plus dbias_part [n_part, 768] fp32: per-column partial sums of the branch-input gradient (dzx with
 * dropout, dz without) = the bias gradient of the dense layer that produced x (BertSelfOutput.dense /
 * BertOutput.dense), which the reference gets from a separate reduction over all rows. */
int t2s_add_layernorm_bwd_bias(const void* dy, const void* z, const float* stats, const float* gamma,
                               void* dz, void* dzx, float* dgamma_part, float* dbeta_part,
                               float* dbias_part, int64_t rows, int dy_dtype, int z_dtype, int dz_dtype,
                               float drop_p, uint64_t drop_seed, t2s_stream_t stream);
/* keep mask (0/1 bytes) of the dropout above for element indices 0..n-1 (test / debugging aid). */
int t2s_dropout_mask(uint8_t* out, int64_t n, float drop_p, uint64_t drop_seed, t2s_stream_t stream);

/* ---- GELU (erf form) of BertIntermediate: y = gelu(u); backward du = dy * gelu'(u) with
 * per-column partial sums of du (bias gradient): dbias_part [n_part, cols] fp32,
 * n_part = t2s_gelu_bwd_parts(rows). */
int t2s_gelu_fwd(const void* u, void* y, int64_t n, int dtype, t2s_stream_t stream);
int t2s_gelu_bwd_parts(int64_t rows);
int t2s_gelu_bwd(const void* dy, const void* u, void* du, float* dbias_part, int64_t rows, int cols,
                 int dtype, t2s_stream_t stream);

/* ---- OCR pointer-network scores (OcrPtrNet.forward t2s.py:648-670 + the concat of
 * T2S._forward_output :279-286): out[b, j, col0 + n] = q[b,j] . k[b,n] * scale + mask01[b,n]
 * (the RAW 0/1 mask is added).  q: [B, D, 768] fp32 (D <= 16 decoding steps), k: [B, N, 768] in
 * k_dtype, mask: [B, N] fp32, out: fp32 rows of out_row_stride elements (the [B, D, V+N] logits
 * buffer, col0 = V).  exact_fp32 != 0 selects the fp32-MFMA path (parity mode, fp32 keys). */
int t2s_ptr_scores(const float* q, const void* k, const float* mask, float* out, int B, int D, int N,
                   int64_t out_row_stride, int col0, float scale, int k_dtype, int exact_fp32,
                   t2s_stream_t stream);

/* ---- grounding scorers and selection (forward only: no gradient reaches them).
 * t2s_question_pool: Grounding_Module._calculate_self_attn t2s.py:453-459 on the already projected
 *   question qp = q_linear(txt_emb) [B, T, 768] fp32: att = softmax(qp.w + bias) over all T, * qmask,
 *   / (sum + 1e-12); out[b] = sum_t att[t] qp[b, t]  -> [B, 768].
 * t2s_attention_score: AttentionScore.forward spatio_temporal_grounding.py:15-23:
 *   score = softmax_M(q . k^T) * mask / (sum + 1e-12), -10000 where mask == 0.  k: [B, M, 768] rows of 768, sample b at
 *   k + b * k_batch_stride elements (>= M * 768: the M rows may be a slice of a longer sequence, e.g. the frame / OCR rows
 *   of the [question; frames; OCR] buffer QTV leaves). */
int t2s_question_pool(const float* qp, const float* w, const float* bias, const float* qmask, float* out,
                      int B, int T, t2s_stream_t stream);
int t2s_attention_score(const float* q, const void* k, int64_t k_batch_stride, const float* mask, float* score,
                        int B, int M, int k_dtype, t2s_stream_t stream);

/* t2s_ground_select: Temporal_Grounding_Indicator.forward (spatio_temporal_grounding.py:34-68),
 * Grounding_Module.forward t2s.py:486-494 and Spatial_Grounding_Indicator.forward (:79-142) in one
 * call.  Inputs: frame_score [B, F] (t2s_attention_score over the frames), frame_mask [B, F] fp32,
 * expo_frame [B, 2, F] / expo_ocr [B, 2, F*P]: the exponential draws of the two gumbel_softmax
 * calls (injected by the caller; noise = -log(expo)), frame_id [B, F] / temporal_id [B, F*P] int64,
 * q_global [B, 768], ocr_feat [B, F*P, 768] (ocr_dtype), bbox [B, F*P, 4].
 * Outputs: pos/neg_obj_mask [B, F], ground_frame [B, frame_topk] int64 (frame ids, ascending frame
 * index), new_ocr_mask / ocr_score / pos_ocr_mask / neg_ocr_mask [B, F*P], ground_box
 * [B, F*ocr_topk, 4].  Ties among equal scores go to the LOWEST index (the reference's ATen order is
 * implementation defined). */
int t2s_ground_select(const float* frame_score, const float* frame_mask, const float* expo_frame,
                      const int64_t* frame_id, const float* q_global, const void* ocr_feat,
                      int64_t ocr_batch_stride, int ocr_dtype,
                      const float* expo_ocr, const int64_t* temporal_id, const float* bbox,
                      float* pos_obj_mask, float* neg_obj_mask, int64_t* ground_frame, float* new_ocr_mask,
                      float* ocr_score, float* pos_ocr_mask, float* neg_ocr_mask, float* ground_box,
                      int B, int F, int P, int frame_topk, int ocr_topk, t2s_stream_t stream);

/* ---- feature-row embedding (T2S._forward_obj_encoding / _forward_ocr_encoding t2s.py:192-258 up to
 * the Linear): out[row] = [ f0[row]/max(||f0[row]||,1e-12) | f1[row]/max(||f1[row]||,1e-12) |
 * emb0[id0[row]] | emb1[id1[row]] ] in out_dtype with row stride ld_out.  f0: [rows, d0] fp32,
 * f1: [rows, d1] fp32 or NULL (d1 = 0), id*: [rows] int64 or NULL, emb*: [emb_rows, emb_dim] fp32.
 * Frames: f0 = video_feat (1024), id0 = frame_id.  OCR: f0 = fastText (300), f1 = PHOC (604),
 * id0 = temporal_id, id1 = track_id. */
int t2s_embed_rows(const float* f0, int d0, const float* f1, int d1, const int64_t* id0, const float* emb0,
                   const int64_t* id1, const float* emb1, int emb_dim, int emb_rows, void* out, int ld_out,
                   int64_t rows, int out_dtype, t2s_stream_t stream);

/* ---- elementwise passes around the encoders (csrc/glue.hip).  Tensors are [B, rows, 768] fp32 unless noted; a pointer paired
 * with a batch stride (elements) may be a slice of a larger buffer.
 * t2s_tanh_residual_fwd: QTV's modality residual y = x + tanh(enc_out) (pythia/models/t2s.py:428-432, the three slices as one
 *   tensor).  t2s_tanh_residual_bwd: g_enc = gy * (1 - tanh(enc_out)^2) in g_dtype (the encoder's output gradient).
 * t2s_add_cast: out = a + b with b in b_dtype (a residual-path gradient meeting the operand-dtype input gradient of an encoder). */
int t2s_tanh_residual_fwd(const float* x, const float* enc_out, float* y, int64_t B, int64_t rows,
                          int64_t y_batch_stride, t2s_stream_t stream);
int t2s_tanh_residual_bwd(const float* gy, int64_t gy_batch_stride, const float* enc_out, void* g_enc, int g_dtype,
                          int64_t B, int64_t rows, t2s_stream_t stream);
int t2s_add_cast(const float* a, int64_t a_batch_stride, const void* b, int b_dtype, float* out, int64_t B,
                 int64_t rows, t2s_stream_t stream);

/* ---- tail of the OCR-token encoding, T2S._forward_ocr_encoding pythia/models/t2s.py:221-258:
 *   out = dropout(LN_feat(a) + LN_bbox(bbox W_box^T + b_box))   with a [rows, 768] = linear_ocr_feat_to_mmt_in(...) (a_dtype),
 * bbox [rows, 4] fp32, W_box [768, 4] / b_box [768] = linear_ocr_bbox_to_mmt_in, the two LayerNorm affines fp32; out [rows, 768]
 * fp32, stats [rows, 4] = (mean_a, rstd_a, mean_b, rstd_b).  Backward: d_a [rows, 768] in a_dtype and partial sums
 * part [t2s_ocr_tail_parts(rows), 9, 768] fp32 = dgamma_a | dbeta_a | dgamma_b | dbeta_b | db_box | dW_box[:, 0..3]
 * (the caller sums over the first axis).  Dropout: the stateless mask of t2s_dropout_mask over element indices row * 768 + col. */
int t2s_ocr_tail_parts(int64_t rows);
int t2s_ocr_tail_fwd(const void* a, int a_dtype, const float* bbox, const float* w_box, const float* b_box,
                     const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b, float* out,
                     float* stats, int64_t rows, float eps, float drop_p, uint64_t drop_seed, t2s_stream_t stream);
int t2s_ocr_tail_bwd(const float* g_out, const void* a, int a_dtype, const float* bbox, const float* w_box,
                     const float* b_box, const float* gamma_a, const float* gamma_b, const float* stats, void* d_a,
                     float* part, int64_t rows, float drop_p, uint64_t drop_seed, t2s_stream_t stream);

/* ---- losses (pythia/modules/losses.py).
 * t2s_bce_masked: POSBCEWithMaskLoss.forward :329-343.  scores/targets/grad: [rows, cols] fp32,
 *   row_mask: [rows]; row_loss[r] = mask[r] * sum_c BCEWithLogits(x, t); grad = (sigmoid(x)-t)*mask[r]
 *   (the caller divides by max(sum(mask), 1)).
 * t2s_infonce_stats / t2s_infonce_bwd: InfoNCE.forward :361-385.  q/p/n = ref/pos/neg logits
 *   [rows, cols] fp32; stats[r] = (q.q, p.p, n.n, q.p, q.n); given gstats = dLoss/dstats the
 *   backward writes dq = 2 g0 q + g3 p + g4 n, dp = 2 g1 p + g3 q, dn = 2 g2 n + g4 q. */
int t2s_bce_masked(const float* scores, const float* targets, const float* row_mask, float* row_loss,
                   float* grad, int64_t rows, int cols, t2s_stream_t stream);
int t2s_infonce_stats(const float* q, const float* p, const float* n, float* stats, int64_t rows, int cols,
                      t2s_stream_t stream);
int t2s_infonce_bwd(const float* q, const float* p, const float* n, const float* gstats, float* dq, float* dp,
                    float* dn, int64_t rows, int cols, t2s_stream_t stream);

/* ---- PHOC descriptor of OCR tokens (the producer of context_feature_1): replaces the reference's C extension
 * pythia/utils/phoc/src/cphoc.c:12-117 as called per token by PhocProcessor (pythia/datasets/processors.py:904-928)
 * through pythia/utils/phoc/build_phoc.py:9-14.  tokens: [n_tokens, width] bytes in HBM, each slot an already
 * normalised token (lower-cased, stripped, only [a-z0-9] kept - build_phoc.py:10-11 stays on the host) padded with
 * NUL; out: [n_tokens, out_row_stride] fp32, the first 604 elements of a row are written with exactly 0.0 / 1.0
 * (14 unigram regions x 36 + 2 bigram regions x 50).  A byte outside [a-z0-9] before the first NUL - where the
 * reference raises RuntimeError - turns that token's row into NaN; the host wrapper rejects such input beforehand. */
int t2s_phoc(const uint8_t* tokens, int64_t n_tokens, int width, float* out, int64_t out_row_stride, t2s_stream_t stream);

/* ---- FastText OCR-token vectors (context_feature_0) from a table resident in HBM.  Replaces the host lookup of
 * FastTextProcessor._map_strings_to_indices pythia/datasets/processors.py:478-491 -> WordToVectorDict pythia/utils/vocab.py:375-381
 * -> third-party fasttext FastText::getWordVector.  table: [table_rows, dim] fp32 = the model's input matrix (nwords + bucket
 * rows); CSR batch: offsets [slots + 1] int32 into ids / word_end; ids [nnz] int32 subword rows of the slot's words in order
 * (-1 = a word with no row); word_end [nnz] uint8 = 1 on the last entry of each word.  out [slots, dim] fp32:
 * mean over words of ((sum of the word's rows) * float(1 / #rows)), in fastText's order of fp32 operations; an empty slot is 0. */
int t2s_fasttext_rows(const float* table, int64_t table_rows, int dim, const int32_t* ids,
                      const uint8_t* word_end, const int32_t* offsets, int64_t slots, float* out,
                      t2s_stream_t stream);

/* ---- global-norm clip + Adam, multi-tensor (BaseTrainer._backward pythia/trainers/base_trainer.py:262-272: clip_gradients
 * pythia/utils/general.py:32-41 = torch.nn.utils.clip_grad_norm_(params, max_norm), then torch.optim.Adam.step(),
 * build_utils.py:54-83; weight_decay 0, no amsgrad).  desc: device table [n_tensors, 5] int64 rows (param, grad, exp_avg,
 * exp_avg_sq pointers - fp32, contiguous - and numel); chunks: device table [n_chunks, 2] int32 rows (tensor, chunk within the
 * tensor) with t2s_optim_chunk_elems() elements per chunk; partials: [n_chunks] fp32 workspace; norm_coef: [2] fp32 = (total
 * L2 norm, clip coefficient min(1, max_norm / (norm + 1e-6))); group_of: device [n_tensors] int32 param-group index; group_lr:
 * HOST array of n_groups <= 8 learning rates; step: 1-based Adam step count (bias corrections); norm_coef may be NULL for
 * t2s_adam_step (no clipping); write_grad != 0 writes the clipped gradients back (the reference clips p.grad in place). */
int t2s_optim_chunk_elems(void);
/* Sticky status of a train step: t2s_status_accumulate ORs uint32 word `word` of a launch's workspace (T2S_FUSED_STATUS_WORD of the
 * fused attention backward's) into sticky[0], counts the launch in sticky[1] and sets sticky[2] / sticky[3] to 1 when bit 0 / bit 1 of
 * the word is set (the bits as flags of their own: a MAX reduction of the four words over ranks then keeps every bit); sticky: 4
 * uint32 of device memory the caller keeps and zeroes.  t2s_status_gate, enqueued between t2s_clip_coef and t2s_adam_step, makes the
 * step a no-op when sticky[0] | sticky[2] | sticky[3] != 0: norm_coef = (NaN, -1) and t2s_adam_step returns without touching
 * parameters, moments or gradients. */
#define T2S_FUSED_STATUS_WORD 24
int t2s_status_accumulate(const void* workspace, int word, void* sticky, t2s_stream_t stream);
int t2s_status_gate(const void* sticky, float* norm_coef, t2s_stream_t stream);
int t2s_grad_sqnorm(const int64_t* desc, const int32_t* chunks, int n_chunks, float* partials, t2s_stream_t stream);
int t2s_clip_coef(const float* partials, int n_chunks, float max_norm, float* norm_coef, t2s_stream_t stream);
int t2s_adam_step(const int64_t* desc, const int32_t* chunks, int n_chunks, const int32_t* group_of,
                  const float* group_lr, int n_groups, float beta1, float beta2, float eps, int step,
                  const float* norm_coef, int write_grad, t2s_stream_t stream);

/* ---- own bf16 MFMA GEMMs of the BERT block's linear layers (round 5, vitxt_gqa_amd/csrc/gemm_bf16.hip) -------------------
 * Replace the torch.nn.Linear calls inside the third-party BertSelfOutput / BertIntermediate / BertOutput the reference runs at
 * pythia/models/t2s.py:423-427, 538-542, 622-626 - and their autograd gradients, stepped by
 * pythia/trainers/base_trainer.py:262-272 - where an epilogue a library GEMM cannot fuse pays: bf16 operands, fp32 accumulation,
 * 256 x 256 x 64 tiles on the 8-phase LDS-DMA pipeline; t2s_gemm_nt launches one PERSISTENT workgroup per CU that
 * walks its XCD's tiles and fetches the next tile's first K-tile under the current tile's epilogue (round 6).
 *
 * t2s_gemm_nt:  C[M, N] = A[M, K] W[N, K]^T, row strides lda / ldw / ldc (elements; multiples of 8), K a multiple of 128, N of 8.
 *   epilogue 0  C = bf16(acc + bias)                  (bias [N] bf16 or NULL)                       nn.Linear forward / dgrad
 *            1  C = bf16(C + bf16(acc))               (the residual branch a LayerNorm backward left in C)
 *            2  C = bf16(acc * gelu'(u[m, n])), colsum_part[2 * ceil(M / 256)][N] fp32 = column sums per 128-row group of the
 *               UNROUNDED fp32 products acc * gelu'(u) (not of the bf16 values stored in C); a non-finite u[m, n] makes its product NaN
 *               (u [M, N] bf16 row stride ldc = the FFN pre-activation; table = fp32 gelu'(x) of every bf16 bit pattern from
 *               t2s_gelu_tables): BertIntermediate's GELU backward + the FFN bias gradient folded into the dgrad of BertOutput.dense
 *            3  C = u = bf16(acc + bias) and g[m, n] = gelu(u) (table = bf16 gelu(x) of every bf16 bit pattern): BertIntermediate
 *               forward with both the pre-activation (kept for backward) and the activation written by the GEMM
 * t2s_gemm_nt_colsum_rows(M): rows of colsum_part for epilogue 2.
 * t2s_gelu_tables: fills fwd_bf16 [65536] bf16 and / or grad_f32 [65536] fp32 with gelu / gelu' (exact erf form, the arithmetic
 *   of t2s_gelu_fwd / t2s_gelu_bwd) of the bf16 value whose bit pattern is the index. */
int t2s_gelu_tables(void* fwd_bf16, void* grad_f32, t2s_stream_t stream);
int t2s_gemm_nt_colsum_rows(int64_t M);
int t2s_gemm_nt(const void* a, const void* w, const void* bias, void* c, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                int64_t ldc, int epilogue, const void* u, void* g, const void* table, float* colsum_part, t2s_stream_t stream);
/* t2s_gemm_wgrad:  dw[n_out, n_in] (fp32; += when accumulate) = dy[rows, n_out]^T x[rows, n_in]  (bf16 operands, row strides
 * ld_dy / ld_x): the weight gradient of a linear layer over all token rows.  The rows are cut into `splits` groups (one
 * 256 x 256 tile x one group per workgroup), each writes an fp32 slab, a second kernel sums the slabs in group order:
 * bit-reproducible.  slabs: splits * n_out * n_in floats of workspace.  n_out, n_in multiples of 256.
 * t2s_gemm_wgrad_splits: the group count that fills the card's CUs once (0: shape not supported). */
int t2s_gemm_wgrad_splits(int64_t rows, int n_out, int n_in);
int t2s_gemm_wgrad(const void* dy, const void* x, float* dw, float* slabs, int64_t rows, int n_out, int n_in, int64_t ld_dy,
                   int64_t ld_x, int splits, int accumulate, t2s_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* T2S_HIP_H */
