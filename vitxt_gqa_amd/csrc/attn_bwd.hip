// Flash-style self-attention backward for gfx950 (MI355X), head_dim 64.
//
// Gradient of the BertSelfAttention block the reference runs eagerly (see attn_fwd.hip for the call
// sites).  P is recomputed from Q, K and the forward's log-sum-exp; nothing of size L x L is stored.
//   delta = rowsum(dO * O);  P = exp(scale*S - LSE);  dS = P * (dO V^T - delta)
//   dV = P^T dO;  dK = scale * dS^T Q;  dQ = scale * dS K
// Two MFMA kernels, both deterministic (no atomics):
//   * dK/dV: key-stationary.  A workgroup = 4 waves = 128 keys of the compacted key list of one
//     (batch, head); each wave keeps dK^T and dV^T of its 32 keys in accumulators (key on the lane)
//     while the workgroup sweeps 32-row query tiles staged in LDS.  S and dP are computed with the
//     key on the MFMA lane, so their accumulators are directly the B operands of the dV^T / dK^T
//     products; the A operands (dO^T, Q^T) are ds_read_b64_tr_b16 reads of the same LDS tiles.
//   * dQ: query-stationary, same skeleton as the forward (query on the lane): S^T and dP^T from row
//     reads of the K and V tiles, dQ^T += K^T dS^T with transposed reads of the K tile.
// The fp32 variants keep the data flow on v_mfma_f32_32x32x2_f32.
#include <type_traits>

#include "attn_common.h"

namespace {

constexpr int BK = 64;
constexpr int TILE64 = 64 * 128;    // bytes of a 64-row bf16 tile

// ---------------------------------------------------------------------------------------------
// delta[b, h, q] = sum_d dO[b, q, h, d] * O[b, q, h, d].  One wave per token row (768 elements).
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ delta,
                                                         int B, int H, int Lq, int64_t o_rs, int64_t o_bs) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * Lq) return;
  const int b = (int)(row / Lq), q = (int)(row % Lq);
  const T* op = o + (int64_t)b * o_bs + (int64_t)q * o_rs;
  const T* dp = dout + (int64_t)b * o_bs + (int64_t)q * o_rs;
  const int nchunk = H * 16;              // 4-element chunks per row
  for (int c0 = 0; c0 < nchunk; c0 += 64) {
    const int ci = c0 + lane;
    float s = 0.f;
    if (ci < nchunk) {
      const f32x4 a = Vec4<T>::load(op + ci * 4), d = Vec4<T>::load(dp + ci * 4);
      s = a[0] * d[0] + a[1] * d[1] + a[2] * d[2] + a[3] * d[3];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if ((lane & 15) == 0 && ci < nchunk) delta[((int64_t)b * H + (ci >> 4)) * Lq + q] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// dQ, bf16.  Query-stationary (query on the lane), 32 query rows per wave, 64-key tiles double-buffered in LDS.
// VALU diet (these kernels are VALU-issue bound at head_dim 64, see attn_fwd_bf16.hip):
//   * Q is pre-scaled by scale*log2(e) and the S accumulators are seeded with -LSE*log2(e) through the MFMA C operand, so
//     P = exp2(S) is ONE v_exp; the dP accumulators are seeded with -delta, so dS = P * dP' is ONE v_mul (with attention
//     dropout the mask has to act before delta is subtracted, so that variant seeds zero and subtracts);
//   * per-lane LDS offsets computed once (4 row-fragment + 4 transposed-fragment addresses), global staging through a
//     uniform base + 32-bit lane offset;
//   * the unmasked tiles [0, nfast) run a loop with no masking code; the edge tiles run the same body with the masks;
//   * the tile body is a wavefront S,dP(block 0) | S,dP(block 1) + softmax(block 0) | dQ(block 0) + softmax(block 1) |
//     dQ(block 1) with scheduling fences between the stages.
template <bool USE_IDX, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_dq_bf16_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE64];   // [buf][K, V]
  __shared__ uint32_t ck_s[2][32];                  // dropout: column keys of the tile's 32 key pairs (packed 16-bit halves)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  int qblk, h, b;
  if (!attn_xcd_tile((p.Lq + 127) / 128, p.H, p.B, qblk, h, b)) return;         // workgroup-uniform
  const int q0 = qblk * 128 + wave * 32;
  const int qrow = q0 + lr;
  const bool qvalid = qrow < p.Lq;
  const int qr = qvalid ? qrow : p.Lq - 1;
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  const int ntiles = (nk + BK - 1) / BK;
  const int nfast = n_prefix / BK;                  // tiles [0, nfast) lie wholly inside the prefix keys: no masks
  const char* __restrict__ K = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.kv_bs + h * 64);
  const char* __restrict__ V = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.kv_bs + h * 64);
  const int32_t* __restrict__ idx = USE_IDX ? p.kv_idx + (int64_t)b * p.idx_cap : nullptr;
  const float c = p.scale * LOG2E;

  bf16x8 qf[4], dof[4];
  {
    const bf16_t* qp = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.q_bs + h * 64 + (int64_t)qr * p.q_rs + 8 * lh;
    const bf16_t* dp = reinterpret_cast<const bf16_t*>(p.dout) + (int64_t)b * p.o_bs + h * 64 + (int64_t)qr * p.o_rs + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
      dof[s] = *reinterpret_cast<const bf16x8*>(dp + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (bf16_t)((float)qf[s][j] * c);
    }
  }
  // accumulator seeds, constant for the whole kernel: -LSE*log2e (rows past Lq: -inf => P = 0) and -delta
  const float del = p.delta[((int64_t)b * p.H + h) * p.Lq + qr];
  f32x16 seed_s, seed_dp;
  {
    const float nl = qvalid ? -p.lse[((int64_t)b * p.H + h) * p.Lq + qr] * LOG2E : -INFINITY;
    const float nd = DROP ? 0.f : -del;
#pragma unroll
    for (int i = 0; i < 16; ++i) { seed_s[i] = nl; seed_dp[i] = nd; }
  }
  const int qdec = qrow - p.dec_q0;
  const uint32_t salt = DROP ? attn_drop_salt(p.drop_seed_lo, p.drop_seed_hi, (uint32_t)(b * p.H + h)) : 0u;
  const uint32_t rh = DROP ? attn_drop_rowhash(salt, qr) : 0u;                   // this lane's row hash; its row key of a tile's key window in
  uint32_t rk2 = 0;                                                              // both 16-bit halves is derived per tile (DQ_ROWKEY)
#define DQ_ROWKEY(t_) if (DROP) rk2 = attn_drop_rowkey16w(rh, ((t_) * BK) / ATTN_DROP_KWIN) * 0x10001u;
  const uint32_t th2 = attn_drop_thresh2k(p.drop_thresh);      // (keep words: attn_drop_pair_kept)
  const float inv_f = p.drop_inv;
  uint32_t ckreg = 0;
  // column keys of key pair `tid` of tile t_ (threads 0..31), staged beside the K/V tile
#define CK_LOAD(t_)                                                                                 \
  if (DROP && tid < 32) {                                                                           \
    const int kp_ = (t_) * BK + 2 * tid;                                                            \
    ckreg = attn_drop_colkey16(salt, kp_, (qblk * 128) / ATTN_DROP_QWIN) | (attn_drop_colkey16(salt, kp_ + 1, (qblk * 128) / ATTN_DROP_QWIN) << 16); \
  }

  int ka[4];                 // row fragment: row lr, chunk 2s + lh
#pragma unroll
  for (int s = 0; s < 4; ++s) ka[s] = tile_off(lr, 2 * s + lh);
  int va[2][2];              // transposed fragment: rows 4lh + qq (+8), chunk 4db + 2g1 + (pp >> 1)
  {
    const int g1 = (lane >> 4) & 1, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int chunk = 4 * db + 2 * g1 + (pp >> 1);
      va[db][0] = tile_off(4 * lh + qq, chunk) + ((pp & 1) << 3);
      va[db][1] = tile_off(4 * lh + qq + 8, chunk) + ((pp & 1) << 3);
    }
  }
#define ROW_FRAG(off_, blk_, s_) (*reinterpret_cast<const bf16x8*>(smem + (ka[s_] + (off_)) + (blk_) * 4096))
  auto tr_frag = [&](const int off, const int rbase, const int db) __attribute__((always_inline)) {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + (va[db][0] + off) + rbase * 128));
    const s16x4 bb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + (va[db][1] + off) + rbase * 128));
    const s16x8 cc = {a[0], a[1], a[2], a[3], bb[0], bb[1], bb[2], bb[3]};
    return __builtin_bit_cast(bf16x8, cc);
  };

  const int sr = tid >> 3, sc = tid & 7;
  uint4 kr0, kr1, vr0, vr1;
  // key-list lookups run one tile ahead of the row loads that depend on them (otherwise every tile waits out a full
  // index-load latency before its K/V loads can even be issued)
  uint32_t ri0, ri1;
#define IDX_LOAD(t_)                                                                                \
  {                                                                                                 \
    int p0_ = (t_) * BK + sr, p1_ = p0_ + 32;                                                       \
    p0_ = p0_ < nk ? p0_ : nk - 1;                                                                  \
    p1_ = p1_ < nk ? p1_ : nk - 1;                                                                  \
    ri0 = USE_IDX ? (uint32_t)idx[p0_] : (uint32_t)p0_;                                             \
    ri1 = USE_IDX ? (uint32_t)idx[p1_] : (uint32_t)p1_;                                             \
  }
#define STAGE_LOAD_ROWS()                                                                           \
  {                                                                                                 \
    const uint32_t o0_ = (ri0 * (uint32_t)p.kv_rs + (uint32_t)sc * 8u) * 2u;                        \
    const uint32_t o1_ = (ri1 * (uint32_t)p.kv_rs + (uint32_t)sc * 8u) * 2u;                        \
    kr0 = *reinterpret_cast<const uint4*>(K + o0_);                                                 \
    vr0 = *reinterpret_cast<const uint4*>(V + o0_);                                                 \
    kr1 = *reinterpret_cast<const uint4*>(K + o1_);                                                 \
    vr1 = *reinterpret_cast<const uint4*>(V + o1_);                                                 \
  }
#define STAGE_LOAD(t_)                                                                              \
  {                                                                                                 \
    IDX_LOAD(t_);                                                                                   \
    STAGE_LOAD_ROWS();                                                                              \
  }
#define STAGE_WRITE(buf_)                                                                           \
  {                                                                                                 \
    char* kb_ = smem + (buf_) * 2 * TILE64 + tile_off(sr, sc);                                      \
    *reinterpret_cast<uint4*>(kb_) = kr0;                                                           \
    *reinterpret_cast<uint4*>(kb_ + TILE64) = vr0;                                                  \
    *reinterpret_cast<uint4*>(kb_ + 4096) = kr1;                                                    \
    *reinterpret_cast<uint4*>(kb_ + TILE64 + 4096) = vr1;                                           \
    if (DROP && tid < 32) ck_s[buf_][tid] = ckreg;                                                  \
  }

  f32x16 dqacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dqacc[0][i] = 0.f; dqacc[1][i] = 0.f; }
  if (ntiles > 0) {
    STAGE_LOAD(0);
    CK_LOAD(0);
    STAGE_WRITE(0);
    IDX_LOAD(1);                                       // indices of tile 1 (clamped), consumed by the first iteration
  }
  __syncthreads();

  // dS of key block kbk_ (P = exp2(S'), dS = P * dP') as bf16 operand fragments
#define SOFTMAX_BLOCK(kbk_, t_, masked_)                                                            \
  {                                                                                                 \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                \
      float pv = fast_exp2(sacc[kbk_][r]);                                                          \
      const int pos = (t_) * BK + (kbk_) * 32 + acc_row(r, lh);                                     \
      if (masked_) {                                                                                \
        const bool ok = pos < nk && (pos < n_prefix || qdec >= pos - n_prefix);                     \
        pv = ok ? pv : 0.f;                                                                         \
      }                                                                                             \
      float dpv = dpacc[kbk_][r];                                                                   \
      if (DROP) {   /* dA = dD * M / (1 - p): registers (r, r+1), r even, are the key pair (kbk*32 + acc_row(r, lh)) / 2;       \
                       M / (1 - p) as a float that is 1/(1-p) or 0: the constant AND the pair's keep half (v_and_b32_sdwa) */ \
        const uint32_t d_ = attn_drop_pair_kept(rk2, ck_s[buf][(kbk_) * 16 + 4 * (r >> 2) + 2 * lh + ((r & 3) >> 1)], th2); \
        const float g_ = (r & 1) ? attn_drop_keep_hi(inv_f, d_) : attn_drop_keep_lo(inv_f, d_);                            \
        dpv = __builtin_fmaf(dpv, g_, -del);                                                        \
      }                                                                                             \
      dpacc[kbk_][r] = pv * dpv;                                                                    \
    }                                                                                               \
    dsf[kbk_][0] = acc_to_frag(dpacc[kbk_], 0);                                                     \
    dsf[kbk_][1] = acc_to_frag(dpacc[kbk_], 1);                                                     \
  }
#define SDP_MFMAS(kb_, vb_, kbk_)                                                                   \
  _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                   \
    sacc[kbk_] = mfma_bf16(ROW_FRAG(kb_, kbk_, s), qf[s], s == 0 ? seed_s : sacc[kbk_]);    /* S'^T[key, q] */  \
    dpacc[kbk_] = mfma_bf16(ROW_FRAG(vb_, kbk_, s), dof[s], s == 0 ? seed_dp : dpacc[kbk_]); /* dP'^T[key, q] */ \
  }
#define DQ_MFMAS(kb_, kbk_)                                                                         \
  _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                     \
  _Pragma("unroll") for (int db = 0; db < 2; ++db)                                                  \
    dqacc[db] = mfma_bf16(tr_frag(kb_, (kbk_) * 32 + 16 * s, db), dsf[kbk_][s], dqacc[db]);   /* dQ^T[d, q] += K^T[d, key] dS^T[key, q] */

  auto run_tiles = [&](auto masked_tag, const int t0, const int t1) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    for (int t = t0; t < t1; ++t) {
      const int buf = t & 1;
      DQ_ROWKEY(t);                                      // row key of this tile's key window
      STAGE_LOAD_ROWS();                                 // tile t+1 (past the end: clamped copies, harmless)
      IDX_LOAD(t + 2);                                   // its indices are not needed before the next iteration
      CK_LOAD(t + 1);
      const int kb = buf * 2 * TILE64, vb = kb + TILE64;
      f32x16 sacc[2], dpacc[2];
      bf16x8 dsf[2][2];
      __builtin_amdgcn_sched_barrier(0);
      SDP_MFMAS(kb, vb, 0);                              // stage A
      __builtin_amdgcn_sched_barrier(0);
      SDP_MFMAS(kb, vb, 1);                              // stage B: 8 MFMA beside the softmax of block 0
      SOFTMAX_BLOCK(0, t, MASKED);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      DQ_MFMAS(kb, 0);                                   // stage C: 4 MFMA beside the softmax of block 1
      SOFTMAX_BLOCK(1, t, MASKED);
      __builtin_amdgcn_sched_barrier(0);
      DQ_MFMAS(kb, 1);                                   // stage D
      STAGE_WRITE(buf ^ 1);
      __syncthreads();
    }
  };
  run_tiles(std::false_type{}, 0, nfast < ntiles ? nfast : ntiles);
  run_tiles(std::true_type{}, nfast < ntiles ? nfast : ntiles, ntiles);
#undef SOFTMAX_BLOCK
#undef SDP_MFMAS
#undef DQ_MFMAS
#undef ROW_FRAG
#undef STAGE_LOAD
#undef STAGE_LOAD_ROWS
#undef IDX_LOAD
#undef CK_LOAD
#undef STAGE_WRITE

  char* ob = smem + wave * (32 * 144);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 t4 = {(bf16_t)(dqacc[db][4 * g] * p.scale), (bf16_t)(dqacc[db][4 * g + 1] * p.scale),
                   (bf16_t)(dqacc[db][4 * g + 2] * p.scale), (bf16_t)(dqacc[db][4 * g + 3] * p.scale)};
      *reinterpret_cast<bf16x4*>(ob + lr * 144 + (db * 32 + 8 * g + 4 * lh) * 2) = t4;
    }
  __syncthreads();
  bf16_t* __restrict__ DQ = reinterpret_cast<bf16_t*>(p.dq) + (int64_t)b * p.q_bs + h * 64;
  bf16_t* __restrict__ DK = reinterpret_cast<bf16_t*>(p.dk) + (int64_t)b * p.kv_bs + h * 64;
  bf16_t* __restrict__ DV = reinterpret_cast<bf16_t*>(p.dv) + (int64_t)b * p.kv_bs + h * 64;
  const uint8_t* __restrict__ rv = p.row_valid ? p.row_valid + (int64_t)b * p.valid_len : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = i * 64 + lane, r = id >> 3, cc = id & 7;
    const int row = q0 + r;
    if (row < p.Lq) {
      *reinterpret_cast<uint4*>(DQ + (int64_t)row * p.q_rs + cc * 8) = *reinterpret_cast<const uint4*>(ob + r * 144 + cc * 16);
      // a row no key list entry points at (a prefix row with row_valid 0, or a row behind the prefix that is not one of this
      // call's n_dec decoder rows - the decoder rows of the OTHER passes in the shared-prefix layout): its dK / dV are exact zeros
      if (rv && (row < p.valid_len ? !rv[row] : (row < p.dec_q0 || row >= p.dec_q0 + p.n_dec))) {
        *reinterpret_cast<uint4*>(DK + (int64_t)row * p.kv_rs + cc * 8) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(DV + (int64_t)row * p.kv_rs + cc * 8) = make_uint4(0, 0, 0, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 variants (parity mode).  Tiles are staged as padded fp32 rows; every MFMA operand is a
// single ds_read_b32 / register value of v_mfma_f32_32x32x2_f32.
constexpr int F32_LD = 65;

template <bool USE_IDX>
__global__ __launch_bounds__(256, 2) void attn_dkdv_f32_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 32 * F32_LD + 64];
  float* qs = smem;
  float* dos = smem + 32 * F32_LD;
  float* lse_s = smem + 2 * 32 * F32_LD;
  float* del_s = lse_s + 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  const int32_t* __restrict__ idx = USE_IDX ? p.kv_idx + (int64_t)b * p.idx_cap : nullptr;
  // the last key block of the grid (sized from the static bound max_keys) continues while the list has more keys: a bound
  // that does not hold costs time, never gradients
  for (int kp0 = blockIdx.x * 128; kp0 < nk; kp0 = (blockIdx.x == gridDim.x - 1) ? kp0 + 128 : 0x3fffff00) {
  const int kpos = kp0 + wave * 32 + lr;
  const bool kvalid = kpos < nk;
  const int kclamp = kvalid ? kpos : nk - 1;
  const int64_t krow = USE_IDX ? (int64_t)idx[kclamp] : (int64_t)kclamp;
  const float* __restrict__ Q = reinterpret_cast<const float*>(p.q) + (int64_t)b * p.q_bs + h * 64;
  const float* __restrict__ DO = reinterpret_cast<const float*>(p.dout) + (int64_t)b * p.o_bs + h * 64;
  const float* __restrict__ LSE = p.lse + ((int64_t)b * p.H + h) * p.Lq;
  const float* __restrict__ DELTA = p.delta + ((int64_t)b * p.H + h) * p.Lq;
  // B operands: B[k = lh][key] = K[key][2t + lh]
  float kf[32], vf[32];
  {
    const float* kp = reinterpret_cast<const float*>(p.k) + (int64_t)b * p.kv_bs + h * 64 + krow * p.kv_rs + lh;
    const float* vp = reinterpret_cast<const float*>(p.v) + (int64_t)b * p.kv_bs + h * 64 + krow * p.kv_rs + lh;
#pragma unroll
    for (int t = 0; t < 32; ++t) { kf[t] = kp[2 * t]; vf[t] = vp[2 * t]; }
  }
  const int kdec = kpos - n_prefix;
  const float c = p.scale * LOG2E;
  const uint32_t salt = attn_drop_salt(p.drop_seed_lo, p.drop_seed_hi, (uint32_t)(b * p.H + h));
  const int nqt = (p.Lq + 31) / 32;
  const int sr = tid >> 3, sc = tid & 7;     // 32 rows x 8 chunks of 8 floats
  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dkacc[0][i] = 0.f; dkacc[1][i] = 0.f; dvacc[0][i] = 0.f; dvacc[1][i] = 0.f; }

  for (int qt = 0; qt < nqt; ++qt) {
    __syncthreads();
    {
      const int r = qt * 32 + sr;
      const bool ok = r < p.Lq;
      const int rc = ok ? r : p.Lq - 1;
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(Q + (int64_t)rc * p.q_rs + sc * 8 + 4 * j2);
        f32x4 d4 = *reinterpret_cast<const f32x4*>(DO + (int64_t)rc * p.o_rs + sc * 8 + 4 * j2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          qs[sr * F32_LD + sc * 8 + 4 * j2 + j] = q4[j];
          dos[sr * F32_LD + sc * 8 + 4 * j2 + j] = ok ? d4[j] : 0.f;
        }
      }
      if (tid < 32) {
        const int r2 = qt * 32 + tid;
        lse_s[tid] = r2 < p.Lq ? LSE[r2] * LOG2E : INFINITY;
        del_s[tid] = r2 < p.Lq ? DELTA[r2] : 0.f;
      }
    }
    __syncthreads();
    f32x16 sacc, dpacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) { sacc[i] = 0.f; dpacc[i] = 0.f; }
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qs[lr * F32_LD + 2 * t + lh], kf[t], sacc, 0, 0, 0);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dos[lr * F32_LD + 2 * t + lh], vf[t], dpacc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = acc_row(r, lh);
      float pv = fast_exp2(sacc[r] * c - lse_s[qi]);
      const int qdec = qt * 32 + qi - p.dec_q0;
      const bool ok = kvalid && (kdec < 0 || qdec >= kdec);
      pv = ok ? pv : 0.f;
      float dpv = dpacc[r];
      bool keep = true;
      if (p.drop_thresh) {
        int qg = qt * 32 + qi;
        qg = qg < p.Lq ? qg : p.Lq - 1;
        keep = attn_drop_keep16(attn_drop_rowkey16(salt, qg, kpos / ATTN_DROP_KWIN), attn_drop_colkey16(salt, kpos, qg / ATTN_DROP_QWIN), p.drop_thresh);
        dpv = keep ? dpv * p.drop_inv : 0.f;
      }
      dpacc[r] = pv * (dpv - del_s[qi]);       // dS uses the UNdropped probability
      sacc[r] = keep ? pv : 0.f;               // dV uses the dropped one (scaled by 1/(1-p) at the end)
    }
    // dV^T[d, key] += dO^T[d, q] P[q, key]:  A[i = d][k = lh] = dO[q = acc_row(r, lh)][d],  B = P register r
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = acc_row(r, lh);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        dvacc[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(dos[qi * F32_LD + db * 32 + lr], sacc[r], dvacc[db], 0, 0, 0);
        dkacc[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(qs[qi * F32_LD + db * 32 + lr], dpacc[r], dkacc[db], 0, 0, 0);
      }
    }
  }
  if (kvalid) {
    float* dkp = reinterpret_cast<float*>(p.dk) + (int64_t)b * p.kv_bs + h * 64 + krow * p.kv_rs;
    float* dvp = reinterpret_cast<float*>(p.dv) + (int64_t)b * p.kv_bs + h * 64 + krow * p.kv_rs;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * lh;
        f32x4 k4 = {dkacc[db][4 * g] * p.scale, dkacc[db][4 * g + 1] * p.scale, dkacc[db][4 * g + 2] * p.scale, dkacc[db][4 * g + 3] * p.scale};
        const float vs_ = p.drop_thresh ? p.drop_inv : 1.f;
        f32x4 v4 = {dvacc[db][4 * g] * vs_, dvacc[db][4 * g + 1] * vs_, dvacc[db][4 * g + 2] * vs_, dvacc[db][4 * g + 3] * vs_};
        *reinterpret_cast<f32x4*>(dkp + d) = k4;
        *reinterpret_cast<f32x4*>(dvp + d) = v4;
      }
  }
  }   // key blocks of this workgroup
}

template <bool USE_IDX>
__global__ __launch_bounds__(256, 2) void attn_dq_f32_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * F32_LD];
  float* ks = smem;
  float* vs = smem + BK * F32_LD;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = q0 + lr;
  const bool qvalid = qrow < p.Lq;
  const int qr = qvalid ? qrow : p.Lq - 1;
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  const int ntiles = (nk + BK - 1) / BK;
  const float* __restrict__ K = reinterpret_cast<const float*>(p.k) + (int64_t)b * p.kv_bs + h * 64;
  const float* __restrict__ V = reinterpret_cast<const float*>(p.v) + (int64_t)b * p.kv_bs + h * 64;
  const int32_t* __restrict__ idx = USE_IDX ? p.kv_idx + (int64_t)b * p.idx_cap : nullptr;
  float qf[32], dof[32];
  {
    const float* qp = reinterpret_cast<const float*>(p.q) + (int64_t)b * p.q_bs + h * 64 + (int64_t)qr * p.q_rs + lh;
    const float* dp = reinterpret_cast<const float*>(p.dout) + (int64_t)b * p.o_bs + h * 64 + (int64_t)qr * p.o_rs + lh;
#pragma unroll
    for (int t = 0; t < 32; ++t) { qf[t] = qp[2 * t]; dof[t] = dp[2 * t]; }
  }
  const float lse2 = qvalid ? p.lse[((int64_t)b * p.H + h) * p.Lq + qr] * LOG2E : INFINITY;
  const float del = p.delta[((int64_t)b * p.H + h) * p.Lq + qr];
  const float c = p.scale * LOG2E;
  const int qdec = qrow - p.dec_q0;
  const uint32_t salt = attn_drop_salt(p.drop_seed_lo, p.drop_seed_hi, (uint32_t)(b * p.H + h));
  const uint32_t rh = attn_drop_rowhash(salt, qr);
  const int sr = tid >> 4, sc = tid & 15;
  f32x16 dqacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dqacc[0][i] = 0.f; dqacc[1][i] = 0.f; }
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
    f32x16 sacc[2], dpacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { sacc[0][i] = 0.f; sacc[1][i] = 0.f; dpacc[0][i] = 0.f; dpacc[1][i] = 0.f; }
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        sacc[kbk] = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[(kbk * 32 + lr) * F32_LD + 2 * s + lh], qf[s], sacc[kbk], 0, 0, 0);
        dpacc[kbk] = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[(kbk * 32 + lr) * F32_LD + 2 * s + lh], dof[s], dpacc[kbk], 0, 0, 0);
      }
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pos = t * BK + kbk * 32 + acc_row(r, lh);
        const bool ok = pos < nk && (pos < n_prefix || qdec >= pos - n_prefix);
        const float pv = ok ? fast_exp2(sacc[kbk][r] * c - lse2) : 0.f;
        float dpv = dpacc[kbk][r];
        if (p.drop_thresh) dpv = attn_drop_keep16(attn_drop_rowkey16w(rh, pos / ATTN_DROP_KWIN), attn_drop_colkey16(salt, pos, qr / ATTN_DROP_QWIN), p.drop_thresh) ? dpv * p.drop_inv : 0.f;
        dpacc[kbk][r] = pv * (dpv - del);
      }
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbk * 32 + acc_row(r, lh);
#pragma unroll
        for (int db = 0; db < 2; ++db)
          dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[key * F32_LD + db * 32 + lr], dpacc[kbk][r], dqacc[db], 0, 0, 0);
      }
  }
  if (qvalid) {
    float* DQ = reinterpret_cast<float*>(p.dq) + (int64_t)b * p.q_bs + h * 64 + (int64_t)qrow * p.q_rs;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 t4 = {dqacc[db][4 * g] * p.scale, dqacc[db][4 * g + 1] * p.scale, dqacc[db][4 * g + 2] * p.scale, dqacc[db][4 * g + 3] * p.scale};
        *reinterpret_cast<f32x4*>(DQ + db * 32 + 8 * g + 4 * lh) = t4;
      }
  }
}

}  // namespace

static int attn_bwd_impl(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                         float* delta, void* dq, void* dk, void* dv, const int32_t* kv_idx, const int32_t* kv_cnt, int B,
                         int H, int Lq, int idx_cap, int n_dec, int dec_q0, int max_keys, int64_t q_row_stride, int64_t q_batch_stride,
                         int64_t kv_row_stride, int64_t kv_batch_stride, int64_t o_row_stride, int64_t o_batch_stride,
                         float scale, int dtype, float drop_p, uint64_t drop_seed, const uint8_t* row_valid, void* fused_ws,
                         int64_t fused_ws_bytes, int fused_handoff, t2s_stream_t stream) {
  T2S_CHECK_ARG(q && k && v && out && dout && lse && delta && dq && dk && dv, "attn_bwd: null pointer");
  T2S_CHECK_ARG(dtype == T2S_F32 || dtype == T2S_BF16, "attn_bwd: bad dtype %d", dtype);
  T2S_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && idx_cap > 0 && n_dec >= 0 && n_dec <= idx_cap, "attn_bwd: bad shape");
  T2S_CHECK_ARG((kv_idx == nullptr) == (kv_cnt == nullptr), "attn_bwd: kv_idx and kv_cnt must both be given or both be NULL");
  const int al = dtype == T2S_BF16 ? 8 : 4;
  T2S_CHECK_ARG(q_row_stride % al == 0 && kv_row_stride % al == 0 && o_row_stride % al == 0 && q_batch_stride % al == 0 &&
                    kv_batch_stride % al == 0 && o_batch_stride % al == 0, "attn_bwd: strides must be multiples of 16 bytes");
  T2S_CHECK_ARG(B <= 65535 && H <= 65535, "attn_bwd: B/H exceed grid limits");
  T2S_CHECK_ARG(max_keys > 0 && max_keys <= idx_cap, "attn_bwd: max_keys %d outside (0, idx_cap=%d]", max_keys, idx_cap);
  {   // the bf16 kernels address a sample's rows with 32-bit byte offsets from a per-sample base
    const int64_t widest = q_row_stride > o_row_stride ? (q_row_stride > kv_row_stride ? q_row_stride : kv_row_stride)
                                                       : (o_row_stride > kv_row_stride ? o_row_stride : kv_row_stride);
    T2S_CHECK_ARG(dtype != T2S_BF16 || ((int64_t)Lq + 128) * widest * 2 < ((int64_t)1 << 32), "attn_bwd: a sample's rows must span < 4 GB");
  }
  AttnParams p = {};
  p.q = q; p.k = k; p.v = v; p.o = out; p.dout = dout; p.lse = const_cast<float*>(lse); p.delta = delta;
  p.dq = dq; p.dk = dk; p.dv = dv; p.kv_idx = kv_idx; p.kv_cnt = kv_cnt;
  p.B = B; p.H = H; p.Lq = Lq; p.idx_cap = idx_cap; p.n_dec = n_dec; p.dec_q0 = dec_q0;
  p.q_rs = q_row_stride; p.q_bs = q_batch_stride; p.kv_rs = kv_row_stride; p.kv_bs = kv_batch_stride;
  p.o_rs = o_row_stride; p.o_bs = o_batch_stride; p.scale = scale;
  p.row_valid = row_valid; p.valid_len = idx_cap - n_dec;
  hipStream_t st = (hipStream_t)stream;
  if (int e = attn_setup_dropout(p, drop_p, drop_seed, st, "attn_bwd")) return e;
  const int64_t rows = (int64_t)B * Lq;
  dim3 gd((unsigned)((rows + 3) / 4)), blk(256);
  dim3 gkv((max_keys + 127) / 128, H, B), gq((Lq + 127) / 128, H, B);
  if (fused_ws) {      // fused 5-product form (attn_bwd_fused_bf16.hip): bf16
    T2S_CHECK_ARG(dtype == T2S_BF16, "attn_bwd_fused: bf16 only");
    T2S_CHECK_ARG(Lq * (int64_t)H * 64 < ((int64_t)1 << 31), "attn_bwd_fused: a sample's fp32 dQ rows must span < 2^31 elements");
    T2S_CHECK_ARG(((int64_t)(Lq + 63) / 64) * 16384 < ((int64_t)1 << 32), "attn_bwd_fused: a (sample, head)'s running dQ sums must span < 4 GB");
    if (int e = launch_attn_bwd_fused_bf16(p, max_keys, fused_ws, (size_t)fused_ws_bytes, fused_handoff, st)) return e;
  } else if (dtype == T2S_BF16) {
    hipLaunchKernelGGL(attn_delta_kernel<bf16_t>, gd, blk, 0, st, (const bf16_t*)out, (const bf16_t*)dout, delta, B, H, Lq, o_row_stride, o_batch_stride);
    launch_attn_dkdv_bf16(p, max_keys, st);
    const dim3 gqx(attn_xcd_grid((Lq + 127) / 128, H, B));       // XCD-aware 1-D grid (attn_common.h)
    if (p.drop_thresh) {
      if (kv_idx) hipLaunchKernelGGL((attn_dq_bf16_kernel<true, true>), gqx, blk, 0, st, p);
      else hipLaunchKernelGGL((attn_dq_bf16_kernel<false, true>), gqx, blk, 0, st, p);
    } else {
      if (kv_idx) hipLaunchKernelGGL((attn_dq_bf16_kernel<true, false>), gqx, blk, 0, st, p);
      else hipLaunchKernelGGL((attn_dq_bf16_kernel<false, false>), gqx, blk, 0, st, p);
    }
  } else {
    hipLaunchKernelGGL(attn_delta_kernel<float>, gd, blk, 0, st, (const float*)out, (const float*)dout, delta, B, H, Lq, o_row_stride, o_batch_stride);
    if (kv_idx) {
      hipLaunchKernelGGL(attn_dkdv_f32_kernel<true>, gkv, blk, 0, st, p);
      hipLaunchKernelGGL(attn_dq_f32_kernel<true>, gq, blk, 0, st, p);
    } else {
      hipLaunchKernelGGL(attn_dkdv_f32_kernel<false>, gkv, blk, 0, st, p);
      hipLaunchKernelGGL(attn_dq_f32_kernel<false>, gq, blk, 0, st, p);
    }
  }
  T2S_CHECK_LAUNCH("attn_bwd");
  return 0;
}

extern "C" int t2s_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                            float* delta, void* dq, void* dk, void* dv, const int32_t* kv_idx, const int32_t* kv_cnt, int B,
                            int H, int Lq, int idx_cap, int n_dec, int dec_q0, int max_keys, int64_t q_row_stride, int64_t q_batch_stride,
                            int64_t kv_row_stride, int64_t kv_batch_stride, int64_t o_row_stride, int64_t o_batch_stride,
                            float scale, int dtype, float drop_p, uint64_t drop_seed, t2s_stream_t stream) {
  return attn_bwd_impl(q, k, v, out, dout, lse, delta, dq, dk, dv, kv_idx, kv_cnt, B, H, Lq, idx_cap, n_dec, dec_q0, max_keys, q_row_stride,
                       q_batch_stride, kv_row_stride, kv_batch_stride, o_row_stride, o_batch_stride, scale, dtype, drop_p, drop_seed,
                       nullptr, nullptr, 0, 0, stream);
}

extern "C" int t2s_attn_bwd_fill(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                                 float* delta, void* dq, void* dk, void* dv, const int32_t* kv_idx, const int32_t* kv_cnt,
                                 const uint8_t* row_valid, int B, int H, int Lq, int idx_cap, int n_dec, int dec_q0, int max_keys,
                                 int64_t q_row_stride, int64_t q_batch_stride, int64_t kv_row_stride, int64_t kv_batch_stride,
                                 int64_t o_row_stride, int64_t o_batch_stride, float scale, int dtype, float drop_p,
                                 uint64_t drop_seed, t2s_stream_t stream) {
  T2S_CHECK_ARG(row_valid && kv_idx, "attn_bwd_fill: row_valid and the key list are required");
  T2S_CHECK_ARG(dtype == T2S_BF16, "attn_bwd_fill: bf16 only (the fp32 kernels leave unlisted rows to the caller)");
  T2S_CHECK_ARG(Lq >= idx_cap && (n_dec == 0 || (dec_q0 >= idx_cap - n_dec && dec_q0 + n_dec <= Lq)),
                "attn_bwd_fill: self-attention layout expected (query rows = prefix rows, then decoder rows; this call's n_dec decoder rows at dec_q0)");
  return attn_bwd_impl(q, k, v, out, dout, lse, delta, dq, dk, dv, kv_idx, kv_cnt, B, H, Lq, idx_cap, n_dec, dec_q0, max_keys, q_row_stride,
                       q_batch_stride, kv_row_stride, kv_batch_stride, o_row_stride, o_batch_stride, scale, dtype, drop_p, drop_seed,
                       row_valid, nullptr, 0, 0, stream);
}

extern "C" int64_t t2s_attn_bwd_fused_workspace_bytes(int B, int H, int Lq) {
  if (B <= 0 || H <= 0 || Lq <= 0) return 0;
  return (int64_t)attn_bwd_fused_workspace_bytes(B, H, Lq);
}

extern "C" int t2s_attn_bwd_fused(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                                  float* delta, void* dq, void* dk, void* dv, void* workspace, int64_t workspace_bytes, int dq_mode,
                                  const int32_t* kv_idx, const int32_t* kv_cnt,
                                  const uint8_t* row_valid, int B, int H, int Lq, int idx_cap, int n_dec, int dec_q0, int max_keys,
                                  int64_t q_row_stride, int64_t q_batch_stride, int64_t kv_row_stride, int64_t kv_batch_stride,
                                  int64_t o_row_stride, int64_t o_batch_stride, float scale, int dtype, float drop_p, uint64_t drop_seed,
                                  t2s_stream_t stream) {
  T2S_CHECK_ARG(workspace, "attn_bwd_fused: the workspace is required");
  T2S_CHECK_ARG(dq_mode == 0 || ((dq_mode & 0xff) == 1 && (dq_mode & ~0x7ff) == 0),
                "attn_bwd_fused: dq_mode %d (0 = fp32 atomics, 1 = ordered hand-off; hand-off bits: 0x200 = write-through running sums, 0x100 / 0x400 = diagnostic modes of the tests)", dq_mode);
  T2S_CHECK_ARG(dtype == T2S_BF16, "attn_bwd_fused: bf16 only");
  T2S_CHECK_ARG(!row_valid || (kv_idx && Lq >= idx_cap && (n_dec == 0 || (dec_q0 >= idx_cap - n_dec && dec_q0 + n_dec <= Lq))),
                "attn_bwd_fused: row_valid needs the key list and the self-attention layout");
  return attn_bwd_impl(q, k, v, out, dout, lse, delta, dq, dk, dv, kv_idx, kv_cnt, B, H, Lq, idx_cap, n_dec, dec_q0, max_keys, q_row_stride,
                       q_batch_stride, kv_row_stride, kv_batch_stride, o_row_stride, o_batch_stride, scale, dtype, drop_p, drop_seed,
                       row_valid, workspace, workspace_bytes, dq_mode, stream);
}
