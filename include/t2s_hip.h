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

#define T2S_ABI_VERSION 1
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
 * lse: [B, H, Lq] fp32, natural-log-sum-exp of the scaled scores (saved for backward). */
int t2s_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
                 const int32_t* kv_idx, const int32_t* kv_cnt,
                 int B, int H, int Lq, int idx_cap, int n_dec, int dec_q0,
                 int64_t q_row_stride, int64_t q_batch_stride,
                 int64_t kv_row_stride, int64_t kv_batch_stride,
                 int64_t o_row_stride, int64_t o_batch_stride,
                 float scale, int dtype, t2s_stream_t stream);

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
                 float scale, int dtype, t2s_stream_t stream);

/* ---- residual + LayerNorm (BertSelfOutput / BertOutput / BertLayerNorm; also
 * t2s.py:87-88,116-117,685-687): z = x + res (res may be NULL); y = (z-mean)/sqrt(var+eps)*g+b,
 * biased variance, eps inside the sqrt.  rows x 768.
 * x has x_dtype (a GEMM output); res, y, z_out have stream_dtype (the residual stream: fp32 in
 * both compute modes, so bf16 rounding never touches the running hidden state).  y_lo (may be
 * NULL) receives a bf16 copy of y for the next GEMM; y may be NULL when only y_lo is wanted.  z_out (may be NULL; may alias x when the
 * two dtypes are equal) keeps the pre-norm sum for backward; stats: [rows, 2] fp32 (mean, rstd),
 * may be NULL.  Supported (x, stream): (f32,f32), (bf16,f32), (bf16,bf16). */
int t2s_add_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                          void* y, void* y_lo, void* z_out, float* stats, int64_t rows, float eps,
                          int x_dtype, int stream_dtype, t2s_stream_t stream);

/* dz = LN backward wrt z (= grad of both x and res); dgamma_part/dbeta_part: [n_part, 768] fp32
 * partial sums (n_part = t2s_layernorm_bwd_parts(rows)); the caller reduces over dim 0.
 * Supported (dy, z, dz) dtypes: (f32,f32,f32), (f32,f32,bf16), (bf16,f32,bf16), (bf16,bf16,bf16). */
int t2s_layernorm_bwd_parts(int64_t rows);
int t2s_add_layernorm_bwd(const void* dy, const void* z, const float* stats, const float* gamma,
                          void* dz, float* dgamma_part, float* dbeta_part, int64_t rows,
                          int dy_dtype, int z_dtype, int dz_dtype, t2s_stream_t stream);

/* ---- GELU (erf form) of BertIntermediate: y = gelu(u); backward du = dy * gelu'(u) with
 * per-column partial sums of du (bias gradient): dbias_part [n_part, cols] fp32,
 * n_part = t2s_gelu_bwd_parts(rows). */
int t2s_gelu_fwd(const void* u, void* y, int64_t n, int dtype, t2s_stream_t stream);
int t2s_gelu_bwd_parts(int64_t rows);
int t2s_gelu_bwd(const void* dy, const void* u, void* du, float* dbias_part, int64_t rows, int cols,
                 int dtype, t2s_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* T2S_HIP_H */
