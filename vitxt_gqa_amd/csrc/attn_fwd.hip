// Flash-style self-attention forward for gfx950 (MI355X), head_dim 64.
//
// Replaces the eager  softmax(Q K^T / 8 + M) V  of the third-party BertSelfAttention that the
// reference calls at pythia/models/t2s.py:423-427 (QTV), :538-542 (TextBert), :622-626 (MMT),
// where M is the materialised [B,1,L,L] 0/-10000 mask of t2s.py:413-419 / :609-618.
//
// Design (MI355X-first):
//   * one workgroup = 4 waves (one per SIMD) = 128 query rows of one (batch, head); each wave owns
//     32 query rows and the whole head_dim; K/V tiles of 64 keys are staged through LDS once per
//     workgroup and double buffered (global->VGPR prefetch of tile t+1 under the MFMAs of tile t);
//   * the key mask is a compacted key list (t2s_compact_keys): masked keys are never loaded, the
//     staging loads gather K/V rows by index (128-B rows = one cache line each);
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16 so a query row lives on ONE lane (column): the row
//     max/sum are in-lane reductions plus one cross-half exchange, and the S^T accumulator is fed
//     straight back as the B operand of O^T += V^T P^T (no LDS round trip for P); V^T fragments come
//     from ds_read_b64_tr_b16 on the row-major V tile;
//   * O is staged through LDS and written as whole 128-B rows.
// The fp32 variant keeps the same data flow on v_mfma_f32_32x32x2_f32 (exact fp32 products).
#include "attn_common.h"

namespace {

constexpr int BQ = 128;   // query rows per workgroup
constexpr int BK = 64;    // keys per tile

// ------------------------------------------------------------------------------------------
// fp32 variant: same data flow on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains).  Used by the
// fp32 parity mode (logits within 1e-3 of the reference); not the throughput path.
//   S^T step t : A[key i][k = lh] = K[key][2t + lh],  B[k = lh][q] = Q[q][2t + lh]
//   O^T step r : B[k = lh][q] = P^T[key = acc_row(r, lh)][q] = S^T accumulator register r,
//                A[d i][k = lh] = V[key = acc_row(r, lh)][d]
constexpr int F32_LD = 65;   // padded fp32 LDS row (conflict-free column reads)

template <bool USE_IDX>
__global__ __launch_bounds__(256, 2) void attn_fwd_f32_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * F32_LD];
  float* ks = smem;
  float* vs = smem + BK * F32_LD;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * BQ + wave * 32;
  const int qrow = q0 + lr;
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  const int ntiles = (nk + BK - 1) / BK;
  const float* __restrict__ Q = reinterpret_cast<const float*>(p.q) + (int64_t)b * p.q_bs + h * 64;
  const float* __restrict__ K = reinterpret_cast<const float*>(p.k) + (int64_t)b * p.kv_bs + h * 64;
  const float* __restrict__ V = reinterpret_cast<const float*>(p.v) + (int64_t)b * p.kv_bs + h * 64;
  const int32_t* __restrict__ idx = USE_IDX ? p.kv_idx + (int64_t)b * p.idx_cap : nullptr;

  float qf[32];
  const int qrc = qrow < p.Lq ? qrow : p.Lq - 1;
  {
    const float* qp = Q + (int64_t)qrc * p.q_rs + lh;
#pragma unroll
    for (int t = 0; t < 32; ++t) qf[t] = qp[2 * t];
  }
  const uint32_t salt = attn_drop_salt(p.drop_seed_lo, p.drop_seed_hi, (uint32_t)(b * p.H + h));
  const uint32_t rh = attn_drop_rowhash(salt, qrc);
  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const float c = p.scale * LOG2E;
  const int qdec = qrow - p.dec_q0;
  const int sr = tid >> 4, sc = tid & 15;    // staging: 16 rows x 16 float4 per pass, 4 passes

  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = sr + 16 * i;
      int pos = t * BK + r;
      pos = pos < nk ? pos : nk - 1;
      const int64_t row = USE_IDX ? (int64_t)idx[pos] : (int64_t)pos;
      const f32x4 kv4 = *reinterpret_cast<const f32x4*>(K + row * p.kv_rs + sc * 4);
      const f32x4 vv4 = *reinterpret_cast<const f32x4*>(V + row * p.kv_rs + sc * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ks[r * F32_LD + sc * 4 + j] = kv4[j];
        vs[r * F32_LD + sc * 4 + j] = vv4[j];
      }
    }
    __syncthreads();

    f32x16 sacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { sacc[0][i] = 0.f; sacc[1][i] = 0.f; }
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int s = 0; s < 32; ++s)
        sacc[kbk] = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[(kbk * 32 + lr) * F32_LD + 2 * s + lh], qf[s], sacc[kbk], 0, 0, 0);

    float mx = -INFINITY;
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pos = t * BK + kbk * 32 + acc_row(r, lh);
        const bool ok = pos < nk && (pos < n_prefix || qdec >= pos - n_prefix);
        const float sv = ok ? sacc[kbk][r] : -INFINITY;
        sacc[kbk][r] = sv;
        mx = fmaxf(mx, sv);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = fast_exp2((m_run - m_use) * c);
    m_run = m_new;
    const float mc = m_use * c;
    float lsum = 0.f;
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = fast_exp2(sacc[kbk][r] * c - mc);
        lsum += pv;                                        // the normaliser sums the UNdropped probabilities
        const int kp_ = t * BK + kbk * 32 + acc_row(r, lh);
        const bool keep = !p.drop_thresh || attn_drop_keep16(attn_drop_rowkey16w(rh, kp_ / ATTN_DROP_KWIN), attn_drop_colkey16(salt, kp_, qrc / ATTN_DROP_QWIN), p.drop_thresh);
        sacc[kbk][r] = keep ? pv : 0.f;
      }
    l_run = l_run * alpha + lsum;
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbk * 32 + acc_row(r, lh);
#pragma unroll
        for (int db = 0; db < 2; ++db)
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[key * F32_LD + db * 32 + lr], sacc[kbk][r], oacc[db], 0, 0, 0);
      }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = (l_tot > 0.f ? 1.f / l_tot : 0.f) * (p.drop_thresh ? p.drop_inv : 1.f);
  float* __restrict__ O = reinterpret_cast<float*>(p.out) + (int64_t)b * p.o_bs + h * 64;
  if (qrow < p.Lq) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 t4 = {oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv, oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv};
        *reinterpret_cast<f32x4*>(O + (int64_t)qrow * p.o_rs + db * 32 + 8 * g + 4 * lh) = t4;
      }
    if (lh == 0) {
      const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
      p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = m_use * p.scale + logf(l_tot);
    }
  }
}

// ------------------------------------------------------------------------------------------
// key compaction: one workgroup per sample; ascending indices of valid prefix rows, then the
// decoder rows.
__global__ __launch_bounds__(256) void compact_keys_kernel(const uint8_t* __restrict__ valid, int32_t* __restrict__ out_idx,
                                                           int32_t* __restrict__ out_cnt, int L, int idx_cap, int n_dec,
                                                           int dec_row0) {
  __shared__ int wsum[4];
  __shared__ int base_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint8_t* vp = valid + (int64_t)b * L;
  int32_t* op = out_idx + (int64_t)b * idx_cap;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int start = 0; start < L; start += 256) {
    const int i = start + tid;
    const int f = (i < L && vp[i] != 0) ? 1 : 0;
    const unsigned long long bal = __ballot(f);
    const int within = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (f) op[off + within] = i;
    __syncthreads();
    if (tid == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  const int n = base_s;
  if (tid < n_dec) op[n + tid] = dec_row0 + tid;
  // pad the remainder with the last valid entry so that clamped tile loads stay in range
  if (tid == 0) out_cnt[b] = n;
}

__global__ __launch_bounds__(256) void attn_drop_mask_kernel(uint8_t* __restrict__ out, uint32_t seed_lo, uint32_t seed_hi, int H, int Lq,
                                                            int Lk, uint32_t thresh) {
  const int b = blockIdx.z, h = blockIdx.y, q = blockIdx.x;
  const uint32_t salt = attn_drop_salt(seed_lo, seed_hi, (uint32_t)(b * H + h));
  const uint32_t rh = attn_drop_rowhash(salt, q);
  for (int k = threadIdx.x; k < Lk; k += 256)
    out[(((int64_t)b * H + h) * Lq + q) * Lk + k] = attn_drop_keep16(attn_drop_rowkey16w(rh, k / ATTN_DROP_KWIN), attn_drop_colkey16(salt, k, q / ATTN_DROP_QWIN), thresh) ? 1 : 0;
}

int check_common(const AttnParams& p, int dtype) {
  T2S_CHECK_ARG(dtype == T2S_F32 || dtype == T2S_BF16, "attn: bad dtype %d", dtype);
  T2S_CHECK_ARG(p.B > 0 && p.H > 0 && p.Lq > 0 && p.idx_cap > 0, "attn: bad shape B=%d H=%d Lq=%d cap=%d", p.B, p.H, p.Lq, p.idx_cap);
  T2S_CHECK_ARG(p.n_dec >= 0 && p.n_dec <= p.idx_cap, "attn: bad n_dec %d", p.n_dec);
  T2S_CHECK_ARG((p.kv_idx == nullptr) == (p.kv_cnt == nullptr), "attn: kv_idx and kv_cnt must both be given or both be NULL");
  const int al = dtype == T2S_BF16 ? 8 : 4;
  T2S_CHECK_ARG(p.q_rs % al == 0 && p.kv_rs % al == 0 && p.o_rs % al == 0 && p.q_bs % al == 0 && p.kv_bs % al == 0 && p.o_bs % al == 0,
                "attn: strides must be multiples of 16 bytes");
  T2S_CHECK_ARG(p.B <= 65535 && p.H <= 65535, "attn: B/H exceed grid limits");
  // the kernels address a sample's K / V rows by 32-bit BYTE offsets (row index x row stride x element size) ...
  const int64_t rows = p.idx_cap > p.Lq ? p.idx_cap : p.Lq;
  const int64_t esz = dtype == T2S_BF16 ? 2 : 4;
  T2S_CHECK_ARG(p.kv_rs > 0 && rows * p.kv_rs * esz < ((int64_t)1 << 32),
                "attn: a sample's K / V rows must span < 4 GB (rows %lld x row stride %lld x %lld bytes)", (long long)rows, (long long)p.kv_rs, (long long)esz);
  // ... which the bf16 kernels build with 24-bit multiplies (__umul24: full rate where v_mul_lo_u32 runs at a quarter)
  T2S_CHECK_ARG(dtype != T2S_BF16 || (p.kv_rs < ((int64_t)1 << 24) && rows < ((int64_t)1 << 24)),
                "attn (bf16): rows %lld and row stride %lld must both be < 2^24 (24-bit multiplies)", (long long)rows, (long long)p.kv_rs);
  return 0;
}

}  // namespace

// shared by the forward / backward entry points: validates the dropout arguments and fills the params.  The mask is a
// stateless function of the seed (attn_common.h); no workspace.
int attn_setup_dropout(AttnParams& p, float drop_p, uint64_t drop_seed, hipStream_t st, const char* who) {
  (void)st;
  p.drop_seed_lo = p.drop_seed_hi = 0;
  p.drop_thresh = 0;
  p.drop_inv = 1.f;
  if (drop_p <= 0.f) return 0;
  T2S_CHECK_ARG(drop_p < 1.f, "%s: dropout probability %f outside [0, 1)", who, drop_p);
  int th = (int)(drop_p * 65536.f + 0.5f);
  th = th < 1 ? 1 : (th > 65535 ? 65535 : th);
  p.drop_thresh = (uint32_t)th;
  p.drop_inv = 65536.f / (65536.f - (float)th);
  p.drop_seed_lo = (uint32_t)drop_seed;
  p.drop_seed_hi = (uint32_t)(drop_seed >> 32);
  return 0;
}

extern "C" int t2s_attn_dropout_mask(uint8_t* out, int B, int H, int Lq, int Lk, float drop_p, uint64_t drop_seed,
                                     t2s_stream_t stream) {
  T2S_CHECK_ARG(out && B > 0 && H > 0 && Lq > 0 && Lk > 0 && B <= 65535 && H <= 65535, "attn_dropout_mask: bad arguments");
  AttnParams p = {};
  p.B = B; p.H = H; p.Lq = Lq;
  if (int e = attn_setup_dropout(p, drop_p, drop_seed, (hipStream_t)stream, "attn_dropout_mask")) return e;
  T2S_CHECK_ARG(p.drop_thresh != 0, "attn_dropout_mask: drop_p must be > 0");
  hipLaunchKernelGGL(attn_drop_mask_kernel, dim3(Lq, H, B), dim3(256), 0, (hipStream_t)stream, out, p.drop_seed_lo, p.drop_seed_hi, H, Lq, Lk,
                     p.drop_thresh);
  T2S_CHECK_LAUNCH("attn_dropout_mask");
  return 0;
}

extern "C" int t2s_compact_keys(const uint8_t* valid, int32_t* out_idx, int32_t* out_cnt, int B, int L, int idx_cap,
                                int n_dec, int dec_row0, t2s_stream_t stream) {
  T2S_CHECK_ARG(valid && out_idx && out_cnt, "compact_keys: null pointer");
  T2S_CHECK_ARG(B > 0 && L > 0 && n_dec >= 0 && n_dec <= 256 && idx_cap >= L + n_dec, "compact_keys: bad shape B=%d L=%d cap=%d n_dec=%d", B, L, idx_cap, n_dec);
  hipLaunchKernelGGL(compact_keys_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, valid, out_idx, out_cnt, L, idx_cap, n_dec, dec_row0);
  T2S_CHECK_LAUNCH("compact_keys");
  return 0;
}

extern "C" int t2s_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, const int32_t* kv_idx,
                            const int32_t* kv_cnt, int B, int H, int Lq, int idx_cap, int n_dec, int dec_q0,
                            int64_t q_row_stride, int64_t q_batch_stride, int64_t kv_row_stride, int64_t kv_batch_stride,
                            int64_t o_row_stride, int64_t o_batch_stride, float scale, int dtype, float drop_p, uint64_t drop_seed,
                            t2s_stream_t stream) {
  T2S_CHECK_ARG(q && k && v && out && lse, "attn_fwd: null pointer");
  AttnParams p = {};
  p.q = q; p.k = k; p.v = v; p.out = out; p.lse = lse; p.kv_idx = kv_idx; p.kv_cnt = kv_cnt;
  p.B = B; p.H = H; p.Lq = Lq; p.idx_cap = idx_cap; p.n_dec = n_dec; p.dec_q0 = dec_q0;
  p.q_rs = q_row_stride; p.q_bs = q_batch_stride; p.kv_rs = kv_row_stride; p.kv_bs = kv_batch_stride;
  p.o_rs = o_row_stride; p.o_bs = o_batch_stride; p.scale = scale;
  if (int e = check_common(p, dtype)) return e;
  dim3 grid((Lq + BQ - 1) / BQ, H, B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (int e = attn_setup_dropout(p, drop_p, drop_seed, st, "attn_fwd")) return e;
  if (dtype == T2S_BF16) {
    launch_attn_fwd_bf16(p, st);
  } else {
    if (kv_idx) hipLaunchKernelGGL(attn_fwd_f32_kernel<true>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(attn_fwd_f32_kernel<false>, grid, block, 0, st, p);
  }
  T2S_CHECK_LAUNCH("attn_fwd");
  return 0;
}
