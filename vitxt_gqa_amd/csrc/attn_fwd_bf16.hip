// bf16 flash-attention forward kernel for gfx950 (see attn_fwd.hip for the design notes and the reference call sites:
// the BERT self-attention of t2s.py:384-432 / 556-633).  Templated on QB = number of 32-row query blocks per wave:
//   QB = 2: a wave owns 64 query rows (256 per workgroup); every K row fragment and V^T fragment read from LDS feeds
//           two MFMAs, and the two blocks give the scheduler independent work to put beside the MFMAs;
//   QB = 1: 32 rows per wave (128 per workgroup) for short sequences.
//
// What bounds this kernel at head_dim 64 is the VALU, not the matrix pipe: every 32x32x16 MFMA (32 cycles) owns two
// S elements per lane, and a classic online softmax spends ~13 VALU instructions on them (max, scale-subtract, exp,
// sum, convert, rescale).  Measured on MI355X (tools/ablate/clock_probe.hip): a wave pair sustains an MFMA every
// ~35 cycles with four v_exp between MFMAs IF the fillers are spread between the MFMAs, while separate MFMA and VALU
// phases simply add up.  So the steady-state tile is built to need ~5 VALU per MFMA and to have them interleaved:
//
//  * Q is pre-scaled by scale*log2(e) (one bf16 rounding per element, once per wave) and kept in LDS.
//  * No running maximum.  The S accumulators are SEEDED with -m (m = the row's reference maximum, the 16-register
//    vector `negm` as the MFMA C operand), so P = exp2(S) is one v_exp per element: no max, no subtract.  m only has
//    to be close enough for exp2 not to overflow, and the row sum certifies that: all P >= 0, so sum < BIG bounds
//    every P.  Softmax is shift-invariant, so any valid m gives the same result.
//  * Tile 0 (which fixes m) and the masked edge tiles run first through a general path (S from zero, masks, true
//    maximum, rescale); softmax does not care about the order of the keys.  The unmasked tiles [1, nfast) then run
//    the steady-state loop, which has no branch and no masking code in it.
//  * A row whose steady-state sum is not < BIG (inf / NaN included; needs a score 80 log2-units above the
//    reference, i.e. e^55 times the largest probability seen so far) poisons the wave: its rows get LSE = NaN and the
//    REPAIR launch (same kernel, every tile through the general path) recomputes exactly those workgroups.
//  * The steady state's row sums run on the MATRIX pipe (round 6): one v_mfma_f32_16x16x32_bf16 with a 0 / 1 A operand per P
//    fragment accumulates the denominator of the lane's own query over all tiles (see `lacc` below): 66 v_add per tile
//    leave the vector-issue port (loop: 502 -> 420 instructions, VALU 339 -> 271), and the sum is over the very rounded
//    probabilities the numerator multiplies.  Measured, same box, interleaved (profiles/r06_fwd_rowsum_ab.txt):
//    7.80 vs 7.97 ms at B = 32, L = 10 132, dropout 0.1 (-2.2 %) with the 1 MFMA : 8 VALU interleave pattern below (1 : 10
//    before: -1.8 %; 12 groups of 1 : 11: -1.6 %).
//  * The tile body is written as a wavefront - S(key block 0) | S(key block 1) + exp(block 0) | PV(block 0) +
//    exp(block 1) | PV(block 1) - with scheduling fences between the stages and an MFMA:VALU interleave pattern
//    inside them.
#include <stdlib.h>

#include "attn_common.h"

namespace {

// FWD_PRIO_MODE (probe builds only, -DFWD_PRIO_MODE=n through tools/ablate/fwd_variant.sh): wave priority by stage of the steady-state tile -
// 1: the MFMA-only stages A and D at s_setprio 1; 2: the softmax-carrying stages B and C at 1; 3: the whole tile at 1 in the waves of every
// second workgroup (the two co-resident workgroups of a CU at different priorities).  0 (the product): no s_setprio; all three measured
// in round 6 (profiles/r06_fwd_prio_ab.txt).
#ifndef FWD_PRIO_MODE
#define FWD_PRIO_MODE 0
#endif
#ifndef FWD_RK_EVERY_TILE
#define FWD_RK_EVERY_TILE 0   // 1: the dropout row key recomputed on every tile (the form until round 6; 7.89 vs 7.83 ms at B = 32: profiles/r06_fwd_prio_ab.txt)
#endif
#ifndef FWD_ABL
#define FWD_ABL 0          // TIMING-ONLY ablations of the steady state (tools/ablate; results wrong): 1 no Q fragment reads, 2 no barrier, 4 no K / V staging, 8 K / V rows loaded and waited for but not written to LDS, 16 LDS writes without the loads, 32 every tile loads the SAME rows (cache-hot loads), 64 no column-key hashing in the steady state
#endif
#define FWD_PRIO(stage_ad_, v_)                                                                     \
  if constexpr ((FWD_PRIO_MODE == 1 && (stage_ad_)) || (FWD_PRIO_MODE == 2 && !(stage_ad_))) __builtin_amdgcn_s_setprio(v_);
constexpr int BK = 64;                     // keys per tile
constexpr int TILE_BYTES = BK * 128;       // one 64-row bf16 tile image
constexpr float BIG = 1.2089258e24f;       // 2^80

// (the experiment forms of this kernel - K / V tiles by LDS-DMA, Q fragments kept in registers, one wave per SIMD, the workgroup
// timeline, word-by-word dropout masks - live in tools/ablate/attn_fwd_bf16_diag.hip, all measured as nulls or losses: tools/ablate/README.md)
template <bool USE_IDX, int QB, bool DROP, bool REPAIR>
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_kernel(AttnParams p) {
  // [buf][K,V] double buffer, then one pre-scaled (32*QB)-row Q tile per wave
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE_BYTES + 4 * QB * 32 * 128];
  __shared__ uint32_t ck_s[2][32];                  // dropout: column keys of the tile's 32 key pairs (packed 16-bit halves)
  constexpr int BQ = 128 * QB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  int qblk, h, b;
  if (!attn_xcd_tile((p.Lq + BQ - 1) / BQ, p.H, p.B, qblk, h, b)) return;       // workgroup-uniform
  const int q0 = qblk * BQ + wave * (32 * QB);
  if (REPAIR) {   // only workgroups holding a poisoned row (LSE = NaN) do anything
    int bad = 0;
    for (int r = tid; r < BQ; r += 256) {
      const int row = qblk * BQ + r;
      if (row < p.Lq) { const float l = p.lse[((int64_t)b * p.H + h) * p.Lq + row]; bad |= !(l == l); }
    }
    if (!__syncthreads_or(bad)) return;
  }
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  const int ntiles = (nk + BK - 1) / BK;
  const int nfast = REPAIR ? 0 : n_prefix / BK;   // tiles [0, nfast) lie wholly inside the prefix keys: no masks
  const bf16_t* __restrict__ Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.q_bs + h * 64;
  const char* __restrict__ K = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.kv_bs + h * 64);
  const char* __restrict__ V = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.kv_bs + h * 64);
  const int32_t* __restrict__ idx = USE_IDX ? p.kv_idx + (int64_t)b * p.idx_cap : nullptr;
  const float c = p.scale * LOG2E;

  // Per-lane LDS byte offsets, computed once.  Everything else about an operand read (key block, k-step pair, buffer,
  // Q block) is a multiple of 16 rows * 128 B that does not disturb the swizzle, i.e. an immediate or a uniform add.
  int ka[4];                 // row fragment (K and Q tiles): row lr, chunk 2s + lh
#pragma unroll
  for (int s = 0; s < 4; ++s) ka[s] = tile_off(lr, 2 * s + lh);
  int va[2][2];              // transposed fragment (V tile): rows 4lh + qq (+8), chunk 4db + 2g1 + (pp >> 1)
  {
    const int g1 = (lane >> 4) & 1, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int chunk = 4 * db + 2 * g1 + (pp >> 1);
      va[db][0] = tile_off(4 * lh + qq, chunk) + ((pp & 1) << 3);
      va[db][1] = tile_off(4 * lh + qq + 8, chunk) + ((pp & 1) << 3);
    }
  }
  const int qoff = 2 * 2 * TILE_BYTES + wave * (QB * 32 * 128);
  // K / Q row fragment (A / B operand of S^T = K Q^T): rows 32*blk + lr of the tile image at uniform byte offset off_
#define ROW_FRAG(off_, blk_, s_) (*reinterpret_cast<const bf16x8*>(smem + (ka[s_] + (off_)) + (blk_) * 4096))
  // V^T fragment (A operand of O^T = V^T P^T) of tile rows rbase .. rbase+15 (a multiple of 16), columns 32*db .. +31;
  // the 8 elements of a lane are rows rbase + 8*(j>>2) + 4*lh + (j&3): the k order of accumulator registers 8s..8s+7
  auto tr_frag = [&](const int off, const int rbase, const int db) __attribute__((always_inline)) {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + (va[db][0] + off) + rbase * 128));
    const s16x4 bb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + (va[db][1] + off) + rbase * 128));
    const s16x8 cc = {a[0], a[1], a[2], a[3], bb[0], bb[1], bb[2], bb[3]};
    return __builtin_bit_cast(bf16x8, cc);
  };

  // Q -> LDS, pre-scaled: lane (q = lr, half lh) owns c * Q[q][16s + 8lh .. +7] and is the only reader of what it wrote
  int qdec[QB];             // decoder step of the lane's query row (negative: not a decoder row)
  uint32_t rk2[QB], rh[QB];    // dropout: the lane's row key of the current key window in both 16-bit halves; its 32-bit row hash
  const uint32_t salt = DROP ? attn_drop_salt(p.drop_seed_lo, p.drop_seed_hi, (uint32_t)(b * p.H + h)) : 0u;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qrow = q0 + qb * 32 + lr;
    const int qr = qrow < p.Lq ? qrow : p.Lq - 1;
    if (DROP) { rh[qb] = attn_drop_rowhash(salt, qr); rk2[qb] = 0; }
    const bf16_t* qp = Q + (int64_t)qr * p.q_rs + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 f = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (bf16_t)((float)f[j] * c);
      *reinterpret_cast<bf16x8*>(smem + (ka[s] + qoff) + qb * 4096) = f;
    }
    qdec[qb] = qrow - p.dec_q0;
  }

  // staging: thread -> (rows sr and sr+32, 16-B chunk sc) of the K and of the V tile.  Named registers, unconditional
  // clamped loads, uniform 64-bit base + 32-bit lane offset (a sample's K/V rows span < 4 GB).
  const int sr = tid >> 3, sc = tid & 7;
  uint4 kr0, kr1, vr0, vr1;
  uint32_t ckreg = 0;
  // dropout: column keys of key pair `tid` of tile t_ (threads 0..31), staged beside the K/V tile
#define CK_LOAD(t_)                                                                                 \
  if (DROP && tid < 32) {                                                                           \
    const int kp_ = (t_) * BK + 2 * tid;                                                            \
    ckreg = attn_drop_colkey16(salt, kp_, (qblk * BQ) / ATTN_DROP_QWIN) | (attn_drop_colkey16(salt, kp_ + 1, (qblk * BQ) / ATTN_DROP_QWIN) << 16); /* (BQ divides the window) */ \
  }
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
    /* (24-bit multiply: full rate where v_mul_lo_u32 runs at a quarter; rows < 2^24 and the row stride < 2^24 are checked at launch) */ \
    const uint32_t o0_ = (__umul24(ri0, (uint32_t)p.kv_rs) + (uint32_t)sc * 8u) * 2u;               \
    const uint32_t o1_ = (__umul24(ri1, (uint32_t)p.kv_rs) + (uint32_t)sc * 8u) * 2u;               \
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
    char* kb_ = smem + (buf_) * 2 * TILE_BYTES + tile_off(sr, sc);                                  \
    *reinterpret_cast<uint4*>(kb_) = kr0;                                                           \
    *reinterpret_cast<uint4*>(kb_ + TILE_BYTES) = vr0;                                              \
    *reinterpret_cast<uint4*>(kb_ + 4096) = kr1;                                                    \
    *reinterpret_cast<uint4*>(kb_ + TILE_BYTES + 4096) = vr1;                                       \
    if (DROP && tid < 32) ck_s[buf_][tid] = ckreg;                                                  \
  }

#define Q_FRAG(qb_, s_) ROW_FRAG(qoff, qb_, s_)
  f32x16 oacc[QB][2], sacc[QB][2], negm[QB];
  bf16x8 pf[QB][2][2];
  float m_run[QB], l_run[QB];
  bool poisoned = false;
  // Row sums on the matrix pipe: in the steady state the denominator of a query row is accumulated by an MFMA with a 0 / 1
  // A operand on the very bf16 P fragment that goes into P.V - v_mfma_f32_16x16x32_bf16, D[m, n] = sum_k A[m, k] B[k, n] with B = the
  // fragment as it stands (lane l: column n = l & 15, k block l >> 4: query (n + 16 (kblock & 1)), key half kblock >> 1) and
  // A[m][kblock] = 1 iff (kblock & 1) == ((m >> 2) & 1): lane l's four result registers (rows 4 (l >> 4) .. + 3 of column l & 15) then all
  // hold the sum over BOTH key halves of the lane's OWN query l & 31.  Replaces 66 v_add per tile on the saturated vector-issue port by
  // 8 four-pass MFMAs on a pipe that is 60 % idle, needs no cross-half exchange, and the denominator sums exactly the rounded
  // probabilities the numerator multiplies.
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  f32x4_t lacc[QB];
  bf16x8 ones_a;
  {
    const bf16_t one = (bf16_t)((((lane >> 4) & 1) == (((lane & 15) >> 2) & 1)) ? 1.f : 0.f);
    ones_a = bf16x8{one, one, one, one, one, one, one, one};
  }
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) lacc[qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[qb][0][i] = 0.f; oacc[qb][1][i] = 0.f; negm[qb][i] = INFINITY; }
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
  }

  // P of (query block, key block, k-step pair s) -> bf16 operand fragment, dropout mask applied
#define PACK_P(qb_, kbk_, s_, cbuf_) PACK_P_(qb_, kbk_, s_, cbuf_, false)
#define PACK_P_(qb_, kbk_, s_, cbuf_, rowsum_)                                                      \
  {                                                                                                 \
    bf16x8 f_ = acc_to_frag(sacc[qb_][kbk_], s_);                                                   \
    if (rowsum_) lacc[qb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_a, f_, lacc[qb_], 0, 0, 0);   /* (the UNdropped probabilities) */ \
    if (DROP) { /* word i of the fragment = key pair kbk*16 + 8s + 4(i>>1) + (i&1) + 2lh of the tile, this lane's query row */ \
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));                                   \
      u32x4 w_ = __builtin_bit_cast(u32x4, f_);                                                     \
      const uint32_t th2_ = attn_drop_thresh2s(p.drop_thresh);                                  \
      {   /* the four mask words stage by stage: no packed instruction right behind the one it depends on */ \
        u32x4 m_;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) m_[i] = attn_drop_kept_mul(rk2[qb_], ck_s[cbuf_][(kbk_) * 16 + 8 * (s_) + 4 * (i >> 1) + (i & 1) + 2 * lh]); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) m_[i] = attn_drop_dropped_sub(m_[i], th2_);   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) m_[i] = attn_drop_kept_mask(m_[i]);          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) w_[i] = attn_drop_apply(w_[i], m_[i]);        \
      }                                                                                             \
      f_ = __builtin_bit_cast(bf16x8, w_);                                                          \
    }                                                                                               \
    pf[qb_][kbk_][s_] = f_;                                                                         \
  }
  // O^T[d, q] += V^T[d, key] P^T[key, q] for key block kbk_ of the tile whose V image is at byte offset vb_
#define PV_MFMAS(vb_, kbk_)                                                                         \
  _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                     \
  _Pragma("unroll") for (int db = 0; db < 2; ++db) {                                                \
    const bf16x8 vf_ = tr_frag(vb_, (kbk_) * 32 + 16 * s, db);                                      \
    _Pragma("unroll") for (int qb = 0; qb < QB; ++qb) oacc[qb][db] = mfma_bf16(vf_, pf[qb][kbk_][s], oacc[qb][db]); \
  }

  // ---- general tiles, nothing pipelined: tile 0 (it fixes the reference maximum) and the edge tiles
  // [max(nfast, 1), ntiles) that need masks
  const int nedge0 = nfast > 1 ? nfast : 1;
  const int ngen = ntiles > 0 ? 1 + (ntiles - nedge0) : 0;
  for (int g = 0; g < ngen; ++g) {
    const int t = g == 0 ? 0 : nedge0 + g - 1;
    if (DROP) {                                         // row keys of this tile's key window (a 64-key tile lies inside one 384-key window)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) rk2[qb] = attn_drop_rowkey16w(rh[qb], (t * BK) / ATTN_DROP_KWIN) * 0x10001u;
    }
    __syncthreads();
    STAGE_LOAD(t);
    CK_LOAD(t);
    STAGE_WRITE(0);
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int i = 0; i < 16; ++i) { sacc[qb][0][i] = 0.f; sacc[qb][1][i] = 0.f; }
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = ROW_FRAG(0, kbk, s);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) sacc[qb][kbk] = mfma_bf16(kf, Q_FRAG(qb, s), sacc[qb][kbk]);
      }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = -INFINITY;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pos = t * BK + kbk * 32 + acc_row(r, lh);
          const bool ok = pos < nk && (pos < n_prefix || qdec[qb] >= pos - n_prefix);
          const float sv = ok ? sacc[qb][kbk][r] : -INFINITY;
          sacc[qb][kbk][r] = sv;
          mx = fmaxf(mx, sv);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qb], mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp2(m_run[qb] - m_use);
      m_run[qb] = m_new;
      float ls = 0.f;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(sacc[qb][kbk][r] - m_use);
          sacc[qb][kbk][r] = pv;
          ls += pv;
        }
      l_run[qb] = l_run[qb] * alpha + ls;
      const float seed = (m_new == -INFINITY) ? INFINITY : -m_new;    // no visible key yet: the steady state poisons
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        oacc[qb][0][i] *= alpha;
        oacc[qb][1][i] *= alpha;
        negm[qb][i] = seed;
      }
    }
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        PACK_P(qb, kbk, 0, 0);
        PACK_P(qb, kbk, 1, 0);
      }
      PV_MFMAS(TILE_BYTES, kbk);
    }
  }

  // ---- steady state: tiles [1, nfast), K/V double-buffered in LDS
  if (nfast > 1) {
    __syncthreads();
    STAGE_LOAD(1);
    CK_LOAD(1);
    STAGE_WRITE(1);
    IDX_LOAD(2);                                         // indices of tile 2 (clamped), consumed by the first iteration
    __syncthreads();
    // exp, row sum and operand fragment of one (query block, key block): 16 v_exp, 16 v_add, 8 v_cvt_pk
#define SOFTMAX_BLOCK(qb_, kbk_, t_)                                                                \
  {                                                                                                 \
    f32x16& sa_ = sacc[qb_][kbk_];                                                                  \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) sa_[j] = fast_exp2(sa_[j]);                      \
    PACK_P_(qb_, kbk_, 0, buf, true);                                                               \
    PACK_P_(qb_, kbk_, 1, buf, true);                                                               \
  }
    if constexpr (FWD_PRIO_MODE == 3) { if (blockIdx.x & 8) __builtin_amdgcn_s_setprio(1); }
#if FWD_ABL & 1       // TIMING-ONLY ablation (results wrong): the steady state's 16 Q fragment reads per tile replaced by ONE register fragment
    const bf16x8 q_abl_ = Q_FRAG(0, 0);
#define Q_FRAG_SS(qb_, s_) q_abl_
#else
#define Q_FRAG_SS(qb_, s_) Q_FRAG(qb_, s_)
#endif
    for (int t = 1; t < nfast; ++t) {
      const int buf = t & 1;
#if !(FWD_ABL & (4 | 16))
      STAGE_LOAD_ROWS();                                 // tile t+1 (after the last fast tile: an unused, harmless load)
#if !(FWD_ABL & 32)
      IDX_LOAD(t + 2);                                   // its indices are not needed before the next iteration
#endif
#endif
#if !(FWD_ABL & 64)
      CK_LOAD(t + 1);
#endif
      const int kb = buf * 2 * TILE_BYTES, vb = kb + TILE_BYTES;
#if FWD_RK_EVERY_TILE
      if (DROP) {                                       // (one multiply + shift + or per query block and tile)
#else
      // the row key changes with the KEY WINDOW (ATTN_DROP_KWIN = 6 tiles): recomputed behind a scalar branch on the window's first tile only
      // (a quarter-rate multiply + two more instructions per query block: 12 of the tile's ~335 vector issue slots)
      if (DROP && (t == 1 || (t * BK) % ATTN_DROP_KWIN == 0)) {
#endif
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) rk2[qb] = attn_drop_rowkey16w(rh[qb], (t * BK) / ATTN_DROP_KWIN) * 0x10001u;
      }
      __builtin_amdgcn_sched_barrier(0);
      FWD_PRIO(true, 1);
      // stage A: S(key block 0), seeded
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = ROW_FRAG(kb, 0, s);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) sacc[qb][0] = mfma_bf16(kf, Q_FRAG_SS(qb, s), s == 0 ? negm[qb] : sacc[qb][0]);
      }
      FWD_PRIO(true, 0);
      __builtin_amdgcn_sched_barrier(0);
      FWD_PRIO(false, 1);
      // stage B: S(key block 1) beside the softmax of key block 0
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = ROW_FRAG(kb, 1, s);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) sacc[qb][1] = mfma_bf16(kf, Q_FRAG_SS(qb, s), s == 0 ? negm[qb] : sacc[qb][1]);
      }
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) SOFTMAX_BLOCK(qb, 0, t);
#pragma unroll
      for (int i = 0; i < 4 * QB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);      // 8 VALU
      }
      __builtin_amdgcn_sched_barrier(0);
      // stage C: PV(key block 0) beside the softmax of key block 1
      PV_MFMAS(vb, 0);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) SOFTMAX_BLOCK(qb, 1, t);
#pragma unroll
      for (int i = 0; i < 4 * QB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
      FWD_PRIO(false, 0);
      __builtin_amdgcn_sched_barrier(0);
      FWD_PRIO(true, 1);
      // stage D: PV(key block 1)
      PV_MFMAS(vb, 1);
      FWD_PRIO(true, 0);
#if FWD_ABL & 8
      asm volatile("" ::"v"(kr0.x ^ vr0.y), "v"(kr1.z ^ vr1.w), "v"(kr0.w ^ kr1.x), "v"(vr0.z ^ vr1.y));
#elif !(FWD_ABL & 4)
      STAGE_WRITE(buf ^ 1);
#endif
#if !(FWD_ABL & 2)
      __syncthreads();
#endif
    }
#undef SOFTMAX_BLOCK
  }
#undef PACK_P
#undef PACK_P_
#undef PV_MFMAS
#undef ROW_FRAG
#undef Q_FRAG
#undef Q_FRAG_SS
#undef STAGE_LOAD
#undef STAGE_LOAD_ROWS
#undef IDX_LOAD
#undef CK_LOAD
#undef STAGE_WRITE

  // ---- epilogue: normalise, stage O through LDS (per-wave 32 x 64 tile, 144-B rows), store whole rows
  __syncthreads();
  // the steady state's row sums (both key halves of the lane's query, see lacc): every P >= 0, so the TOTAL < BIG bounds every tile's
  // sum and every single P, as the per-tile test did; inf / NaN included
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) poisoned |= !(lacc[qb][0] < BIG);
  const bool wave_poisoned = __any(poisoned);     // both lane halves of a row, and simplest: the whole wave
  char* ob = smem + wave * (32 * 144);
  bf16_t* __restrict__ O = reinterpret_cast<bf16_t*>(p.out) + (int64_t)b * p.o_bs + h * 64;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qrow = q0 + qb * 32 + lr;
    const float l_tot = (l_run[qb] + __shfl_xor(l_run[qb], 32, 64)) + lacc[qb][0];      // general tiles (per half) + steady state (whole row)
    const float inv = (l_tot > 0.f ? 1.f / l_tot : 0.f) * (DROP ? p.drop_inv : 1.f);   // normaliser uses the UNdropped sum
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 t4 = {(bf16_t)(oacc[qb][db][4 * g] * inv), (bf16_t)(oacc[qb][db][4 * g + 1] * inv),
                     (bf16_t)(oacc[qb][db][4 * g + 2] * inv), (bf16_t)(oacc[qb][db][4 * g + 3] * inv)};
        const int d = db * 32 + 8 * g + 4 * lh;
        *reinterpret_cast<bf16x4*>(ob + lr * 144 + d * 2) = t4;
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = i * 64 + lane, r = id >> 3, cc = id & 7;
      const int row = q0 + qb * 32 + r;
      if (row < p.Lq)
        *reinterpret_cast<uint4*>(O + (int64_t)row * p.o_rs + cc * 8) = *reinterpret_cast<const uint4*>(ob + r * 144 + cc * 16);
    }
    if (lh == 0 && qrow < p.Lq) {
      const float m_use = (m_run[qb] == -INFINITY) ? 0.f : m_run[qb];
      // m is in log2 units; NaN = "recompute this row" for the repair launch
      p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] =
          (wave_poisoned ? __builtin_nanf("") : m_use * 0.6931471805599453f + logf(l_tot));
    }
    if (qb + 1 < QB) __syncthreads();
  }
}

template <bool DROP, bool REPAIR>
void launch_fwd(const AttnParams& p, hipStream_t st) {
  // 64 rows per wave once there is more than one workgroup of queries.  (T2S_ATTN_FWD_QB1=1, probe runs: 32 rows per wave at every length -
  // 157 registers and 48 KB of LDS per workgroup = THREE workgroups per CU instead of two, every K / V fragment feeding one MFMA instead of two:
  // measured in round 6, profiles/r06_fwd_qb1_ab.txt)
  static const bool force_qb1 = [] { const char* e = getenv("T2S_ATTN_FWD_QB1"); return e && atoi(e) != 0; }();
  const bool wide = p.Lq > 256 && !force_qb1;
  dim3 block(256);
  if (wide) {
    dim3 grid(attn_xcd_grid((p.Lq + 255) / 256, p.H, p.B));
    if (p.kv_idx) hipLaunchKernelGGL((attn_fwd_bf16_kernel<true, 2, DROP, REPAIR>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((attn_fwd_bf16_kernel<false, 2, DROP, REPAIR>), grid, block, 0, st, p);
  } else {
    dim3 grid(attn_xcd_grid((p.Lq + 127) / 128, p.H, p.B));
    if (p.kv_idx) hipLaunchKernelGGL((attn_fwd_bf16_kernel<true, 1, DROP, REPAIR>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((attn_fwd_bf16_kernel<false, 1, DROP, REPAIR>), grid, block, 0, st, p);
  }
}

}  // namespace

void launch_attn_fwd_bf16(const AttnParams& p, hipStream_t st) {
  // main pass (the one-wave-per-SIMD experiment of round 3, 5-12 % slower on the benchmark shape, is tools/ablate/attn_fwd_pw_bf16.hip)
  if (p.drop_thresh) launch_fwd<true, false>(p, st);
  else launch_fwd<false, false>(p, st);
  // The steady-state loop exists only when some sample can have more than one whole tile of prefix keys; only then can
  // a wave have poisoned its rows.  The repair launch reads the LSE of its rows and returns unless one is NaN.
  if ((p.idx_cap - p.n_dec) / BK > 1) {
    if (p.drop_thresh) launch_fwd<true, true>(p, st);
    else launch_fwd<false, true>(p, st);
  }
}

