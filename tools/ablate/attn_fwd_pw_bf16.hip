// bf16 flash-attention forward for gfx950, head_dim 64, ONE wave per SIMD with a hand-placed instruction stream
// (the BERT self-attention of pythia/models/t2s.py:384-432 / 556-633; same arithmetic, masks, dropout function and outputs as
// attn_fwd_bf16.hip, which stays the kernel of short sequences and of the repair launch).
//
// Why a second forward kernel.  At head_dim 64 a 32x32x16 MFMA serves two scores per lane and the softmax spends 5 (no dropout) to
// 9 (dropout) VALU instructions on them: the loop is VALU-issue-bound.  The two-waves-per-SIMD kernel leaves the interleaving of
// MFMA and VALU work to the compiler's scheduler and to the arbitration between two waves that run the same program: 39 % / 51 %
// MFMA-busy (with / without dropout, profiles/mfma_busy.json).  The fused backward showed what a single wave with a hand-placed
// stream does for exactly this chain (S -> exp2 -> packed P -> the MFMA that consumes it): its "G1 + E" slots run 8 MFMAs in 295 /
// 450 cycles (tools/fused_stamps.py).  This kernel is that structure for the forward:
//   * a workgroup = 4 waves, one per SIMD, each wave 64 query rows (two 32-row blocks qb) and the whole 512-register file:
//     O^T accumulators (64 AGPRs), the pre-scaled Q fragments and the seed vectors in registers - Q never touches LDS;
//   * steady-state tiles of 128 keys = 8 blocks b = (key block kb = b >> 1, query block qb = b & 1) of 8 MFMAs each:
//       G1(b) = S^T[kb][qb] = K[kb] Q^T[qb]   (4 MFMAs, the accumulator seeded with -m through the C operand: P = exp2(S) needs no max),
//       E(b)  = P = exp2(S), row sums, bf16 operand words, dropout mask   (8 chunks of 2 scores pairs),
//       G2(b) = O^T[qb] += V^T[kb] P^T   (4 MFMAs);
//     ten slots per tile:  slot k = { G1(b_k), G2(b_{k-2}) } with E(b_{k-1}) chunked between its MFMAs, one MFMA per scheduling
//     fence group.  Every consumer stands at least four MFMAs behind the MFMA (or the VALU instruction) that produces its input -
//     the asm MFMAs get no hazard padding from the compiler (cdna_hip_programming.md section 5.7);
//   * K / V tiles double-buffered in LDS in the tile_off image (attn_common.h), gathered through the key list by register
//     staging, one barrier per 128 keys; LDS loads of a slot's successor are issued at its head, the dropout column keys first
//     (LDS returns in order: the wait that guards them must not drag the fragment loads with it);
//   * tile 0 (it fixes the reference maximum m) and the masked edge tiles run first through the general 64-key path of
//     attn_fwd_bf16.hip (S from zero, masks, true maximum, rescale); a steady-state tile whose row sum is not < 2^80 poisons the wave
//     (LSE = NaN) and the repair launch of attn_fwd_bf16.hip recomputes that workgroup.
#include "attn_common.h"

namespace {

constexpr int GK = 64;                     // keys per general tile
constexpr int FK = 128;                    // keys per steady-state tile
constexpr int FT = FK * 128;               // bytes of a 128-row tile image
constexpr float PW_BIG = 1.2089258e24f;    // 2^80
constexpr int PW_SMEM = 2 * 2 * FT + 2 * 64 * 4;      // [buf][K | V] + dropout column-key words [buf][64]

// PW_ABL (tools/ablate/pw_ablate.sh; never set in a product build): timing experiments that REMOVE one ingredient of the steady
// state - 1 the softmax chunks, 2 the MFMAs, 4 the LDS fragment loads, 8 the global -> LDS staging, 16 the tile barrier.  Results
// are wrong by construction; only the launch duration means anything.
#ifndef PW_ABL
#define PW_ABL 0
#endif

typedef uint32_t pw_u32x4 __attribute__((ext_vector_type(4)));
#define PW_U4(x) __builtin_bit_cast(pw_u32x4, x)
#define PW_FENCE() __builtin_amdgcn_sched_barrier(0)
// S chain: first MFMA takes the seed vector as C, the others accumulate; O chain accumulates in AGPRs
// (the Q fragment, B operand, comes from the accumulator file: an MFMA A / B operand may be an AGPR)
#if PW_ABL & 2
#define PW_MFMA_S0(acc, a, qa, c) asm volatile("" : "=v"(acc) : "v"(PW_U4(a)), "a"(qa), "v"(c))
#define PW_MFMA_S(acc, a, qa) asm volatile("" : "+v"(acc) : "v"(PW_U4(a)), "a"(qa))
#define PW_MFMA_O(acc, a, b) asm volatile("" : "+a"(acc) : "v"(PW_U4(a)), "v"(b))
#else
#define PW_MFMA_S0(acc, a, qa, c) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(PW_U4(a)), "a"(qa), "v"(c))
#define PW_MFMA_S(acc, a, qa) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(PW_U4(a)), "a"(qa))
#define PW_MFMA_O(acc, a, b) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(PW_U4(a)), "v"(b))
#endif

__device__ __forceinline__ uint32_t pw_pack2(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, bf16x2_t{(__bf16)a, (__bf16)b});
}

// transposed fragment (attn_common.h lds_tr_frag) from a tile-row base and this lane's two precomputed offsets
__device__ __forceinline__ bf16x8 pw_tr(const char* base, const int o0, const int o1) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + o0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + o1));
  const s16x8 c = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, c);
}

// chunk M (accumulator registers 2M, 2M + 1 = two consecutive keys of this lane's query) of the softmax of one block
template <int M, bool DROP>
__device__ __forceinline__ void pw_e(f32x16& sacc, uint32_t (&pfw)[8], float& lsum, const uint32_t rk2, const uint32_t ck2, const uint32_t th2) {
#if PW_ABL & 1
  asm volatile("" : "=v"(pfw[M]) : "v"(sacc[2 * M]), "v"(sacc[2 * M + 1]), "v"(ck2));
  return;
#endif
  const float p0 = fast_exp2(sacc[2 * M]), p1 = fast_exp2(sacc[2 * M + 1]);
  lsum += p0;                                          // the normaliser uses the UNdropped probabilities
  lsum += p1;
  asm volatile("" : "+v"(lsum));                       // keep the two adds HERE: left alone the compiler sinks all 128 row-sum adds of a
                                                       // tile behind its last MFMA (128 live registers, 500 cycles beside nothing)
  const uint32_t w = pw_pack2(p0, p1);
  pfw[M] = DROP ? attn_drop_apply(w, attn_drop_pair_dropped(rk2, ck2, th2)) : w;
}

template <bool USE_IDX, bool DROP>
__global__ __launch_bounds__(256, 1) void attn_fwd_pw_bf16_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* const ck_s = reinterpret_cast<uint32_t*>(smem + 2 * 2 * FT);      // [buf][64]: column keys of the tile's 64 key pairs
  constexpr int BQ = 256;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  int qblk, h, b;
  if (!attn_xcd_tile((p.Lq + BQ - 1) / BQ, p.H, p.B, qblk, h, b)) return;       // workgroup-uniform
  const int q0 = qblk * BQ + wave * 64;
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  const int ntiles = (nk + GK - 1) / GK;           // 64-key tiles
  const int nfast = n_prefix / GK;                 // tiles [0, nfast) lie wholly inside the prefix keys: no masks
  const int nsup = nfast > 1 ? (nfast - 1) / 2 : 0;      // steady-state 128-key tiles: 64-key tiles [1, 1 + 2 nsup)
  const bf16_t* __restrict__ Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.q_bs + h * 64;
  const char* __restrict__ K = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.kv_bs + h * 64);
  const char* __restrict__ V = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.kv_bs + h * 64);
  const int32_t* __restrict__ idx = USE_IDX ? p.kv_idx + (int64_t)b * p.idx_cap : nullptr;
  const float c = p.scale * LOG2E;

  // per-lane LDS byte offsets, computed once (attn_fwd_bf16.hip): row fragment of row lr (chunk 2s + lh); transposed fragment
  int ka[4], va[2][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) ka[s] = tile_off(lr, 2 * s + lh);
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
#define TR_FRAG(off_, rbase_, db_) pw_tr(smem + (off_) + (rbase_) * 128, va[db_][0], va[db_][1])

  // Q fragments in registers, pre-scaled by scale * log2(e): lane (q = lr, half lh) owns c * Q[q][16s + 8lh .. +7]
  bf16x8 qreg[2][4];
  int qdec[2];
  uint32_t rk2[2];
  const uint32_t salt = DROP ? attn_drop_salt(p.drop_seed_lo, p.drop_seed_hi, (uint32_t)(b * p.H + h)) : 0u;
  const uint32_t th2 = attn_drop_thresh2s(p.drop_thresh);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = q0 + qb * 32 + lr;
    const int qr = qrow < p.Lq ? qrow : p.Lq - 1;
    rk2[qb] = DROP ? attn_drop_rowkey16(salt, qr, 0) * 0x10001u : 0u;      // (launch_attn_fwd_pw_bf16 declines dropout launches since round 4: the
                                                                           //  row key changes per 384-key window, which this pipeline does not do)
    const bf16_t* qp = Q + (int64_t)qr * p.q_rs + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 f = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (bf16_t)((float)f[j] * c);
      qreg[qb][s] = f;
    }
    qdec[qb] = qrow - p.dec_q0;
  }

  // ---- staging.  General 64-key tile: thread -> rows sr, sr + 32, chunk sc; steady 128-key tile: rows sr + 32 i, i = 0..3.
  const int sr = tid >> 3, sc = tid & 7;
  // (named registers: as arrays these went to scratch - the general loop and the steady loop both write them)
  uint4 kr0, kr1, kr2, kr3, vr0, vr1, vr2, vr3;
  uint32_t ri0, ri1, ri2, ri3;
  uint32_t ckreg = 0;
#define PW_G_LD1(kr_, vr_, i_)                                                                      \
  {                                                                                                 \
    int pp_ = gt_ * GK + sr + 32 * (i_);                                                            \
    pp_ = pp_ < nk ? pp_ : nk - 1;                                                                  \
    const uint32_t r_ = USE_IDX ? (uint32_t)idx[pp_] : (uint32_t)pp_;                               \
    const uint32_t o_ = (r_ * (uint32_t)p.kv_rs + (uint32_t)sc * 8u) * 2u;                          \
    kr_ = *reinterpret_cast<const uint4*>(K + o_);                                                  \
    vr_ = *reinterpret_cast<const uint4*>(V + o_);                                                  \
  }
#define G_STAGE(t_)      /* global -> registers of general tile t_ (64 keys at positions t_ * 64 ..) */        \
  {                                                                                                 \
    const int gt_ = (t_);                                                                           \
    PW_G_LD1(kr0, vr0, 0); PW_G_LD1(kr1, vr1, 1);                                                   \
    if (DROP && tid < 32) {                                                                         \
      const int kp_ = gt_ * GK + 2 * tid;                                                           \
      ckreg = attn_drop_colkey16(salt, kp_, 0) | (attn_drop_colkey16(salt, kp_ + 1, 0) << 16);            \
    }                                                                                               \
  }
#define G_WRITE()        /* into buffer 0: K image at 0, V image at FT (first 64 rows of each) */                \
  {                                                                                                 \
    *reinterpret_cast<uint4*>(smem + tile_off(sr, sc)) = kr0;                                       \
    *reinterpret_cast<uint4*>(smem + FT + tile_off(sr, sc)) = vr0;                                  \
    *reinterpret_cast<uint4*>(smem + tile_off(sr + 32, sc)) = kr1;                                  \
    *reinterpret_cast<uint4*>(smem + FT + tile_off(sr + 32, sc)) = vr1;                             \
    if (DROP && tid < 32) ck_s[tid] = ckreg;                                                        \
  }
#define PW_F_IX1(ri_, i_)                                                                           \
  {                                                                                                 \
    int pp_ = GK + fu_ * FK + sr + 32 * (i_);                                                       \
    pp_ = pp_ < nk ? pp_ : nk - 1;                                                                  \
    ri_ = USE_IDX ? (uint32_t)idx[pp_] : (uint32_t)pp_;                                             \
  }
#define F_IDX(u_)        /* key-list lookups of steady tile u_ (positions 64 + 128 u_ ..), one tile ahead of the row loads */ \
  { const int fu_ = (u_); PW_F_IX1(ri0, 0); PW_F_IX1(ri1, 1); PW_F_IX1(ri2, 2); PW_F_IX1(ri3, 3); }
#define PW_F_LD1(kr_, vr_, ri_)                                                                     \
  {                                                                                                 \
    const uint32_t o_ = (ri_ * (uint32_t)p.kv_rs + (uint32_t)sc * 8u) * 2u;                         \
    kr_ = *reinterpret_cast<const uint4*>(K + o_);                                                  \
    vr_ = *reinterpret_cast<const uint4*>(V + o_);                                                  \
  }
#define F_ROWS() { PW_F_LD1(kr0, vr0, ri0); PW_F_LD1(kr1, vr1, ri1); PW_F_LD1(kr2, vr2, ri2); PW_F_LD1(kr3, vr3, ri3); }
#define F_CK(u_)         /* column keys of the 64 key pairs of steady tile u_: threads 0..63 */                  \
  if (DROP && tid < 64) {                                                                           \
    const int kp_ = GK + (u_) * FK + 2 * tid;                                                       \
    ckreg = attn_drop_colkey16(salt, kp_, 0) | (attn_drop_colkey16(salt, kp_ + 1, 0) << 16);              \
  }
#define PW_F_WR1(kr_, vr_, i_)                                                                      \
  *reinterpret_cast<uint4*>(kb_ + tile_off(sr + 32 * (i_), sc)) = kr_;                              \
  *reinterpret_cast<uint4*>(kb_ + FT + tile_off(sr + 32 * (i_), sc)) = vr_;
#define F_WRITE(buf_)                                                                               \
  {                                                                                                 \
    char* kb_ = smem + (buf_) * 2 * FT;                                                             \
    PW_F_WR1(kr0, vr0, 0); PW_F_WR1(kr1, vr1, 1); PW_F_WR1(kr2, vr2, 2); PW_F_WR1(kr3, vr3, 3);     \
    if (DROP && tid < 64) ck_s[(buf_) * 64 + tid] = ckreg;                                          \
  }

  f32x16 oacc[2][2], negm[2];
  float m_run[2], l_run[2];
  bool poisoned = false;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[qb][0][i] = 0.f; oacc[qb][1][i] = 0.f; negm[qb][i] = INFINITY; }
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
  }

  // ---- general tiles, nothing pipelined: tile 0 (it fixes the reference maximum) and every tile the steady state does not
  // take: [1 + 2 nsup, ntiles) (a left-over unmasked tile when nfast - 1 is odd, then the masked edge tiles)
  const int gen1 = 1 + 2 * nsup;
  const int ngen = ntiles > 0 ? 1 + (ntiles > gen1 ? ntiles - gen1 : 0) : 0;
  for (int g = 0; g < ngen; ++g) {
    const int t = g == 0 ? 0 : gen1 + g - 1;
    __syncthreads();
    G_STAGE(t);
    G_WRITE();
    __syncthreads();
    f32x16 sacc[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int i = 0; i < 16; ++i) { sacc[qb][0][i] = 0.f; sacc[qb][1][i] = 0.f; }
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = ROW_FRAG(0, kbk, s);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) sacc[qb][kbk] = mfma_bf16(kf, qreg[qb][s], sacc[qb][kbk]);
      }
    bf16x8 pf[2][2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float mx = -INFINITY;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pos = t * GK + kbk * 32 + acc_row(r, lh);
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
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 f_ = acc_to_frag(sacc[qb][kbk], s);
          if (DROP) {   // word i of the fragment = key pair kbk*16 + 8s + 4(i>>1) + (i&1) + 2lh of the tile
            pw_u32x4 w_ = PW_U4(f_);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              w_[i] = attn_drop_apply(w_[i], attn_drop_pair_dropped(rk2[qb], ck_s[kbk * 16 + 8 * s + 4 * (i >> 1) + (i & 1) + 2 * lh], th2));
            f_ = __builtin_bit_cast(bf16x8, w_);
          }
          pf[qb][kbk][s] = f_;
        }
    }
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 vf_ = TR_FRAG(FT, kbk * 32 + 16 * s, db);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) oacc[qb][db] = mfma_bf16(vf_, pf[qb][kbk][s], oacc[qb][db]);
        }
  }

  // ---- steady state: 128-key tiles u = 0 .. nsup - 1 (key positions 64 + 128 u ..), K / V double-buffered in LDS.
  // The block pipeline runs ACROSS tiles: slot k = { G1(b_k), G2(b_{k-2}), E(b_{k-1}) } with the block index counting on through the
  // tile boundary, so every slot but the first two of tile 0 and the last two of the last tile has 8 MFMAs beside one block's
  // softmax.  Slots 2..7 of a tile read its LDS images; behind slot 7 nothing of the tile is read any more (the last K / V^T
  // fragments are in registers): there the next tile is written into the other buffer, the ONE barrier of the tile follows, and
  // slots 8 / 9 (= slots 0 / 1 of the next tile) already take its first K fragments.
  if (nsup > 0) {
    __syncthreads();
    F_IDX(0);
    F_ROWS();
    F_CK(0);
    F_WRITE(0);
    F_IDX(1);                                              // (clamped) indices of tile 1
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int db = 0; db < 2; ++db) asm volatile("" : "+a"(oacc[qb][db]));      // the O^T accumulators live in AGPRs from here on
    pw_u32x4 qa[2][4];                                     // ... and so do the Q fragments
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qa[qb][s] = PW_U4(qreg[qb][s]);
        asm volatile("" : "+a"(qa[qb][s]));
      }
    float lsum[2] = {0.f, 0.f};                            // row sums of ALL steady tiles; certified once behind the loop
    f32x16 sacc[2];
    uint32_t pfw[2][8], ckw[8];
    bf16x8 kf[2][4], vf[2][2][2];
    // LDS loads: the column keys of a key block (first in a group: LDS returns in order, the wait that guards their use must not
    // drag the fragment loads with it), the K row fragments, the V^T fragments; image base / key pointer given by the caller
#define PW_LD_CK(ckb_, kb_)                                                                         \
  if (DROP && !(PW_ABL & 4)) { _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                       \
    const uint2 c2_ = *reinterpret_cast<const uint2*>((ckb_) + (kb_) * 16 + 4 * j);                 \
    ckw[2 * j] = c2_.x; ckw[2 * j + 1] = c2_.y; } }
#define PW_LD_KF(kbase_, kb_) if (!(PW_ABL & 4)) _Pragma("unroll") for (int s = 0; s < 4; ++s) kf[(kb_) & 1][s] = ROW_FRAG(kbase_, kb_, s);
#define PW_LD_VF(vbase_, kb_)                                                                       \
  if (!(PW_ABL & 4)) _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                     \
  _Pragma("unroll") for (int db = 0; db < 2; ++db) vf[(kb_) & 1][s][db] = TR_FRAG(vbase_, (kb_) * 32 + 16 * s, db);
    // MFMA m (0..3) of G1(block k): the S^T chain of (kb = k >> 1, qb = k & 1)
#define PW_G1(k_, m_)                                                                               \
  { if ((m_) == 0) PW_MFMA_S0(sacc[(k_) & 1], kf[((k_) >> 1) & 1][0], qa[(k_) & 1][0], negm[(k_) & 1]);      \
    else PW_MFMA_S(sacc[(k_) & 1], kf[((k_) >> 1) & 1][m_], qa[(k_) & 1][m_]); }
    // MFMA m (0..3) of G2(block k): O^T[qb][db] += V^T[kb][s][db] P^T[k][s], (s, db) = (m >> 1, m & 1)
#define PW_G2(k_, m_)                                                                               \
  { const pw_u32x4 b_ = {pfw[(k_) & 1][4 * ((m_) >> 1)], pfw[(k_) & 1][4 * ((m_) >> 1) + 1], pfw[(k_) & 1][4 * ((m_) >> 1) + 2],  \
                         pfw[(k_) & 1][4 * ((m_) >> 1) + 3]};                                        \
    PW_MFMA_O(oacc[(k_) & 1][(m_) & 1], vf[((k_) >> 1) & 1][(m_) >> 1][(m_) & 1], b_); }
#define PW_E(k_, m_) pw_e<m_, DROP>(sacc[(k_) & 1], pfw[(k_) & 1], lsum[(k_) & 1], rk2[(k_) & 1], ckw[m_], th2)
    // slot k, 3 <= k <= 7: G1(b_k) then G2(b_{k-2}), one chunk of E(b_{k-1}) behind each MFMA.  (E(b_{k-1}) reads the chain that
    // ended FOUR MFMAs before the slot: an 8-pass MFMA's result needs 12 wait states before VALU code may read it, and nothing
    // pads asm MFMAs.)
#define PW_SLOT(k_)                                                                                 \
  PW_G1(k_, 0); PW_E((k_) - 1, 0); PW_FENCE(); PW_G1(k_, 1); PW_E((k_) - 1, 1); PW_FENCE();          \
  PW_G1(k_, 2); PW_E((k_) - 1, 2); PW_FENCE(); PW_G1(k_, 3); PW_E((k_) - 1, 3); PW_FENCE();          \
  PW_G2((k_) - 2, 0); PW_E((k_) - 1, 4); PW_FENCE(); PW_G2((k_) - 2, 1); PW_E((k_) - 1, 5); PW_FENCE(); \
  PW_G2((k_) - 2, 2); PW_E((k_) - 1, 6); PW_FENCE(); PW_G2((k_) - 2, 3); PW_E((k_) - 1, 7); PW_FENCE();
    // ---- prologue: rows of tile 1 on their way, slots 0 and 1 of tile 0 (nothing to pair G1(b0) with yet)
    F_ROWS();
    F_IDX(2);
    F_CK(1);
    {
      const uint32_t* ckb = ck_s + 2 * lh;
      PW_LD_KF(0, 0); PW_FENCE();
      PW_G1(0, 0); PW_LD_CK(ckb, 0); PW_LD_KF(0, 1); PW_LD_VF(FT, 0); PW_FENCE(); PW_G1(0, 1); PW_FENCE(); PW_G1(0, 2); PW_FENCE(); PW_G1(0, 3); PW_FENCE();
      // E(b0) reads the chain that ended with the LAST MFMA before this slot: it starts behind the slot's second MFMA
      PW_G1(1, 0); PW_FENCE(); PW_G1(1, 1); PW_E(0, 0); PW_E(0, 1); PW_E(0, 2); PW_FENCE();
      PW_G1(1, 2); PW_E(0, 3); PW_E(0, 4); PW_E(0, 5); PW_FENCE(); PW_G1(1, 3); PW_E(0, 6); PW_E(0, 7); PW_FENCE();
    }
    // slots 2 .. 7 of a tile whose images lie at kbase_ / vbase_ (column keys at ckb_).  Slot 2: G1(b2), G2(b0) + E(b1); G1(b1)
    // ended with the last MFMA before the slot: E(b1) starts behind the second MFMA.  (b1 shares the key block, hence the column
    // keys, of b0.)
#define PW_SLOTS_2_7(kbase_, vbase_, ckb_)                                                          \
  PW_G1(2, 0); PW_FENCE(); PW_G1(2, 1); PW_E(1, 0); PW_FENCE(); PW_G1(2, 2); PW_E(1, 1); PW_FENCE(); PW_G1(2, 3); PW_E(1, 2); PW_FENCE(); \
  PW_G2(0, 0); PW_E(1, 3); PW_FENCE(); PW_G2(0, 1); PW_E(1, 4); PW_FENCE(); PW_G2(0, 2); PW_E(1, 5); PW_E(1, 6); PW_FENCE(); \
  PW_G2(0, 3); PW_E(1, 7); PW_FENCE();                                                              \
  PW_LD_CK(ckb_, 1); PW_LD_KF(kbase_, 2); PW_LD_VF(vbase_, 1); PW_FENCE();                          \
  PW_SLOT(3);                                                                                       \
  PW_SLOT(4);                                                                                       \
  PW_LD_CK(ckb_, 2); PW_LD_KF(kbase_, 3); PW_LD_VF(vbase_, 2); PW_FENCE();                          \
  PW_SLOT(5);                                                                                       \
  PW_SLOT(6);                                                                                       \
  PW_LD_CK(ckb_, 3); PW_LD_VF(vbase_, 3); PW_FENCE();                                               \
  PW_SLOT(7);
    // every tile but the last.  (The last tile is NOT a branch inside this loop: the compiler gave the two arms different
    // accumulator registers and copied them at the join - straight behind asm MFMAs whose result latency it does not know.)
    for (int u = 0; u + 1 < nsup; ++u) {
      const int buf = u & 1;
      const int kbase = buf * 2 * FT, vbase = kbase + FT;
      const uint32_t* ckb = ck_s + buf * 64 + 2 * lh;
      PW_SLOTS_2_7(kbase, vbase, ckb);
      // the next tile goes into the other buffer (last read a whole tile ago; every wave's LDS reads have returned by the barrier
      // of that tile), the tile's one barrier, the rows of the tile after next on their way
      if (!(PW_ABL & 8)) F_WRITE(buf ^ 1);
      if (!(PW_ABL & 16)) __syncthreads();
      if (!(PW_ABL & 8)) {
        F_ROWS();
        F_IDX(u + 3);
        F_CK(u + 2);
      }
      const int nkbase = (buf ^ 1) * 2 * FT, nvbase = nkbase + FT;
      const uint32_t* nckb = ck_s + (buf ^ 1) * 64 + 2 * lh;
      // slot 8 = slot 0 of the next tile: G2(b6) first (its operands are in registers: it covers the latency of the K fragment
      // loads), then G1(b0'); E(b7) beside both
      PW_LD_KF(nkbase, 0); PW_LD_KF(nkbase, 1); PW_LD_VF(nvbase, 0); PW_FENCE();
      PW_G2(6, 0); PW_E(7, 0); PW_FENCE(); PW_G2(6, 1); PW_E(7, 1); PW_FENCE(); PW_G2(6, 2); PW_E(7, 2); PW_FENCE(); PW_G2(6, 3); PW_E(7, 3); PW_FENCE();
      PW_G1(0, 0); PW_E(7, 4); PW_FENCE(); PW_G1(0, 1); PW_E(7, 5); PW_FENCE(); PW_G1(0, 2); PW_E(7, 6); PW_FENCE(); PW_G1(0, 3); PW_E(7, 7); PW_FENCE();
      // slot 9 = slot 1 of the next tile: G2(b7), G1(b1') + E(b0') (behind the second MFMA: G1(b0') has only just ended)
      PW_LD_CK(nckb, 0); PW_FENCE();
      PW_G2(7, 0); PW_FENCE(); PW_G2(7, 1); PW_E(0, 0); PW_FENCE(); PW_G2(7, 2); PW_E(0, 1); PW_FENCE(); PW_G2(7, 3); PW_E(0, 2); PW_FENCE();
      PW_G1(1, 0); PW_E(0, 3); PW_FENCE(); PW_G1(1, 1); PW_E(0, 4); PW_FENCE(); PW_G1(1, 2); PW_E(0, 5); PW_E(0, 6); PW_FENCE();
      PW_G1(1, 3); PW_E(0, 7); PW_FENCE();
    }
    {
      // the last tile: slots 2 .. 7, then slot 8 = G2(b6) + E(b7), slot 9 = G2(b7)
      const int buf = (nsup - 1) & 1;
      const int kbase = buf * 2 * FT, vbase = kbase + FT;
      const uint32_t* ckb = ck_s + buf * 64 + 2 * lh;
      PW_SLOTS_2_7(kbase, vbase, ckb);
      PW_G2(6, 0); PW_E(7, 0); PW_E(7, 1); PW_FENCE(); PW_G2(6, 1); PW_E(7, 2); PW_E(7, 3); PW_FENCE();
      PW_G2(6, 2); PW_E(7, 4); PW_E(7, 5); PW_FENCE(); PW_G2(6, 3); PW_E(7, 6); PW_E(7, 7); PW_FENCE();
      PW_G2(7, 0); PW_FENCE(); PW_G2(7, 1); PW_FENCE(); PW_G2(7, 2); PW_FENCE(); PW_G2(7, 3); PW_FENCE();
    }
#undef PW_SLOTS_2_7
#undef PW_SLOT
#undef PW_E
#undef PW_G2
#undef PW_G1
#undef PW_LD_VF
#undef PW_LD_KF
#undef PW_LD_CK
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      l_run[qb] += lsum[qb];
      if (!PW_ABL) poisoned |= !(lsum[qb] < PW_BIG);                   // every P >= 0: a sum below 2^80 bounds every term of every tile (NaN / inf fail it)
    }
    // the O^T accumulators were last written by asm MFMAs the compiler does not see as such: cover MFMA result -> v_accvgpr_read
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(oacc[qb][0]), "+a"(oacc[qb][1]));
  }
#undef ROW_FRAG
#undef TR_FRAG
#undef G_STAGE
#undef G_WRITE
#undef F_IDX
#undef F_ROWS
#undef F_CK
#undef F_WRITE

  // ---- epilogue: normalise, stage O through LDS (per-wave 32 x 64 tile, 144-B rows), store whole rows
  __syncthreads();
  const bool wave_poisoned = __any(poisoned);
  char* ob = smem + wave * (32 * 144);
  bf16_t* __restrict__ O = reinterpret_cast<bf16_t*>(p.out) + (int64_t)b * p.o_bs + h * 64;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = q0 + qb * 32 + lr;
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
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
      p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = (wave_poisoned ? __builtin_nanf("") : m_use * 0.6931471805599453f + logf(l_tot));
    }
    if (qb + 1 < 2) __syncthreads();
  }
}

}  // namespace

// Main pass of the forward for sequences of more than one 256-row workgroup (the repair launch stays attn_fwd_bf16.hip's).
int launch_attn_fwd_pw_bf16(const AttnParams& p, hipStream_t st) {
  const void* kernels[] = {reinterpret_cast<const void*>(&attn_fwd_pw_bf16_kernel<true, false>), reinterpret_cast<const void*>(&attn_fwd_pw_bf16_kernel<true, true>),
                           reinterpret_cast<const void*>(&attn_fwd_pw_bf16_kernel<false, false>), reinterpret_cast<const void*>(&attn_fwd_pw_bf16_kernel<false, true>)};
  for (const void* k : kernels)
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, PW_SMEM) != hipSuccess) {
      t2s_set_error("attn_fwd: cannot reserve %d bytes of LDS per workgroup", PW_SMEM);
      return 3;
    }
  dim3 grid(attn_xcd_grid((p.Lq + 255) / 256, p.H, p.B)), block(256);
  if (p.kv_idx) {
    if (p.drop_thresh) hipLaunchKernelGGL((attn_fwd_pw_bf16_kernel<true, true>), grid, block, PW_SMEM, st, p);
    else hipLaunchKernelGGL((attn_fwd_pw_bf16_kernel<true, false>), grid, block, PW_SMEM, st, p);
  } else {
    if (p.drop_thresh) hipLaunchKernelGGL((attn_fwd_pw_bf16_kernel<false, true>), grid, block, PW_SMEM, st, p);
    else hipLaunchKernelGGL((attn_fwd_pw_bf16_kernel<false, false>), grid, block, PW_SMEM, st, p);
  }
  return 0;
}
