// bf16 flash-attention forward for long sequences on gfx950: the "ping-pong" form of attn_fwd_bf16.hip
// (same math, same LDS tile image, same reference call sites: the BERT self-attention of t2s.py:384-432 / 556-633).
//
// Why a second form.  On a CDNA4 SIMD the matrix pipe and the VALU issue port are separate, but two co-resident
// waves running the SAME phase order (QK^T MFMAs -> softmax VALU -> PV MFMAs) fall into lock step: both want the
// matrix pipe, then both want the VALU, and the measured tile time is the SUM of the two (attn_fwd_bf16_kernel:
// 33 % MFMA busy).  Here a 512-thread workgroup holds 8 waves = 2 per SIMD, split into two groups that are kept
// half an iteration apart by the workgroup barrier itself:
//
//      phase p      0        1        2        3        4
//      group X    V(0)     M(0)     V(1)     M(1)     V(2)  ...        V(t) = softmax of tile t (VALU: exp, sum, cvt)
//      group Y     -       V(0)     M(0)     V(1)     M(1)  ...        M(t) = O += P(t) V(t);  S(t+1) = K(t+1) Q^T (MFMA)
//
// so on every SIMD one wave is in its MFMA phase while its partner is in its VALU phase.  The loop is software
// pipelined by one tile (S(t+1) is produced in M(t)), K/V tiles live in a 3-deep LDS ring: tile j is read in phases
// 2j-1 .. 2j+2, its global loads are issued by every thread at the start of phase 2j-3 and written at the end of
// phase 2j-2 (two phases of latency cover), and it replaces tile j-3, last read in phase 2j-4.
//
// Softmax without a running maximum in the steady state: Q is pre-scaled by scale*log2(e), the S accumulators are
// seeded with -m (m = the row's reference maximum), so P = exp2(S) is ONE v_exp per
// element with no subtract and no max.  m only has to be close enough for exp2 not to overflow, which the row sum
// certifies (all P >= 0, so sum < BIG bounds every P).  A tile that fails the test (the first tile of a row, a late
// outlier) and the masked edge tiles are redone by the general path inside the same V phase: S from zero, masks,
// true maximum, rescale of O and l, reseed of negm.  Softmax is shift-invariant: same function either way.
#include "attn_common.h"

// The workgroup barrier separates the phases in time; the scheduling barriers keep the compiler from moving register-only
// work (exp, cvt, MFMA) of one phase across it into the other.
#define PHASE_BARRIER()                   \
  {                                       \
    __builtin_amdgcn_sched_barrier(0);    \
    __syncthreads();                      \
    __builtin_amdgcn_sched_barrier(0);    \
  }

#ifndef PP_ROLE
#define PP_ROLE(wave) ((wave) >> 2)        // waves w and w+4 share a SIMD
#endif

namespace {

constexpr int BK = 64;                     // keys per tile
constexpr int TILE_BYTES = BK * 128;       // one 64-row bf16 tile
constexpr int NBUF = 3;
constexpr float BIG = 1.0995116e12f;       // 2^40: P stays far inside the bf16 / fp32 range

template <bool USE_IDX>
__global__ __launch_bounds__(512, 2) void attn_fwd_pp_bf16_kernel(AttnParams p) {
  constexpr bool DROP = false;         // the experiment has no dropout path
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [buf][K,V] ring, then one pre-scaled 64-row Q tile per wave
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int role = PP_ROLE(wave);
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 512 + wave * 64;
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  const int ntiles = (nk + BK - 1) / BK;
  const int nfast = n_prefix / BK;          // tiles [0, nfast) lie wholly inside the prefix keys: no masks
  const bf16_t* __restrict__ Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* __restrict__ K = reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.kv_bs + h * 64;
  const bf16_t* __restrict__ V = reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.kv_bs + h * 64;
  const int32_t* __restrict__ idx = USE_IDX ? p.kv_idx + (int64_t)b * p.idx_cap : nullptr;
  const float c = p.scale * LOG2E;

  // Per-lane LDS byte offsets, computed once.  Everything else about an operand read (key block, k-step pair, ring
  // slot, Q block) is a multiple of 128 rows * 128 B that does not disturb the swizzle, i.e. an immediate or a
  // uniform add: 12 address registers serve all 40 fragment reads of a tile.
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
  const int qoff = NBUF * 2 * TILE_BYTES + wave * (64 * 128);
  // K / Q row fragment: rows 32*blk + lr of the tile image at byte offset off_ (uniform), k-step s
#define ROW_FRAG(off_, blk_, s_) (*reinterpret_cast<const bf16x8*>(smem + (ka[s_] + (off_)) + (blk_) * 4096))
  // V^T fragment of tile rows rbase_ .. rbase_+15 (rbase_ a multiple of 16), columns 32*db .. +31
  auto tr_frag = [&](const int off, const int rbase, const int db) __attribute__((always_inline)) {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + (va[db][0] + off) + rbase * 128));
    const s16x4 bb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + (va[db][1] + off) + rbase * 128));
    const s16x8 cc = {a[0], a[1], a[2], a[3], bb[0], bb[1], bb[2], bb[3]};
    return __builtin_bit_cast(bf16x8, cc);
  };

  // Q fragments (B operand of S^T = K Q^T), pre-scaled: lane (q = lr, half lh) holds c * Q[q][16s + 8lh .. +7].  They
  // live in LDS (each lane re-reads exactly the 16-byte chunks it wrote), not in 32 registers: the M phase already
  // holds O (64), S (64), the seeds (32) and the staged tile.
  int qdec[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = q0 + qb * 32 + lr;
    const int qr = qrow < p.Lq ? qrow : p.Lq - 1;
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

  // staging: thread -> (row sr, 16-B chunk sc) of the K and of the V tile; named registers, unconditional clamped loads
  // (row and chunk are re-derived from a laundered thread index at each use: cheaper than carrying the offsets through
  // the loop, where the allocator would spill them)
  uint4 kr, vr;
#define STAGE_LOAD(j_)                                                                              \
  {                                                                                                 \
    int tid_ = tid;                                                                                 \
    asm volatile("" : "+v"(tid_));                                                                  \
    int p_ = (j_) * BK + (tid_ >> 3);                                                               \
    p_ = p_ < nk ? p_ : nk - 1;                                                                     \
    const uint32_t r_ = USE_IDX ? (uint32_t)idx[p_] : (uint32_t)p_;                                 \
    /* uniform 64-bit base + 32-bit lane offset (one address register per load; a sample's K/V rows span < 4 GB) */ \
    const uint32_t o_ = (r_ * (uint32_t)p.kv_rs + (uint32_t)(tid_ & 7) * 8u) * 2u;                  \
    kr = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(K) + o_);                    \
    vr = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(V) + o_);                    \
  }
#define STAGE_WRITE(slot_)                                                                          \
  {                                                                                                 \
    int tid_ = tid;                                                                                 \
    asm volatile("" : "+v"(tid_));                                                                  \
    char* kb_ = smem + (slot_) * 2 * TILE_BYTES + tile_off(tid_ >> 3, tid_ & 7);                    \
    *reinterpret_cast<uint4*>(kb_) = kr;                                                            \
    *reinterpret_cast<uint4*>(kb_ + TILE_BYTES) = vr;                                               \
  }

  f32x16 oacc[2][2], sacc[2][2];
  float negm[2];      // accumulator seed -m of the lane's query row (INFINITY while the row has no visible key)
  bf16x8 pf[2][2][2];
  float m_run[2], l_run[2];
  bool poisoned = false;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      oacc[qb][0][i] = 0.f; oacc[qb][1][i] = 0.f;
      sacc[qb][0][i] = 0.f; sacc[qb][1][i] = 0.f;
    }
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
    negm[qb] = INFINITY;
  }

  // bf16 operand fragments of P (in sacc), with the dropout mask applied
#define PACK_P(t_)                                                                                  \
  _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                                  \
  _Pragma("unroll") for (int kbk = 0; kbk < 2; ++kbk)                                               \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                   \
    bf16x8 f = acc_to_frag(sacc[qb][kbk], s);                                                       \
    pf[qb][kbk][s] = f;                                                                             \
  }
  // O^T[d, q] += V^T[d, key] P^T[key, q] with V in tile image vb_
#define PV_MFMAS(vb_)                                                                               \
  _Pragma("unroll") for (int kbk = 0; kbk < 2; ++kbk)                                               \
  _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                     \
  _Pragma("unroll") for (int db = 0; db < 2; ++db) {                                                \
    const bf16x8 vf = tr_frag(vb_, kbk * 32 + 16 * s, db);                                \
    _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) oacc[qb][db] = mfma_bf16(vf, pf[qb][kbk][s], oacc[qb][db]); \
  }

#define SEED_S()                                                                                    \
  _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                                  \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                  \
    float s0_, s1_;                                                                                 \
    asm volatile("v_mov_b32 %0, %1" : "=v"(s0_) : "v"(negm[qb]));                                   \
    asm volatile("v_mov_b32 %0, %1" : "=v"(s1_) : "v"(negm[qb]));                                   \
    sacc[qb][0][i] = s0_;                                                                           \
    sacc[qb][1][i] = s1_;                                                                           \
  }

  // ---- general tiles, all waves in step, nothing pipelined: tile 0 (it fixes the reference maximum) and the edge
  // tiles [max(nfast, 1), ntiles) that need masks.  Softmax does not care about the order of the keys.
  const int nedge0 = nfast > 1 ? nfast : 1;
  const int ngen = ntiles > 0 ? 1 + (ntiles - nedge0) : 0;
  for (int g = 0; g < ngen; ++g) {
    const int t = g == 0 ? 0 : nedge0 + g - 1;
    __syncthreads();
    STAGE_LOAD(t);
    STAGE_WRITE(0);
    __syncthreads();
    const int vb = TILE_BYTES;
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
        for (int qb = 0; qb < 2; ++qb) sacc[qb][kbk] = mfma_bf16(kf, ROW_FRAG(qoff, qb, s), sacc[qb][kbk]);
      }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
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
      negm[qb] = (m_new == -INFINITY) ? INFINITY : -m_new;    // no visible key yet: the steady state poisons
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        oacc[qb][0][i] *= alpha;
        oacc[qb][1][i] *= alpha;
      }
    }
    PACK_P(t);
    PV_MFMAS(vb);
  }

  // ---- steady state: tiles [1, nfast)
  if (nfast > 1) {
    __syncthreads();
    STAGE_LOAD(1);
    STAGE_WRITE(1);
    if (nfast > 2) {
      STAGE_LOAD(2);
      STAGE_WRITE(2);
    }
    __syncthreads();
    {   // S(1), seeded
      const int kb = 1 * 2 * TILE_BYTES;
      SEED_S();
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 kf = ROW_FRAG(kb, kbk, s);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) sacc[qb][kbk] = mfma_bf16(kf, ROW_FRAG(qoff, qb, s), sacc[qb][kbk]);
        }
    }

    // V phase of tile t: P(t) = exp2(S(t)), row sums, bf16 operand fragments.  A row whose sum is not < BIG (inf and
    // NaN included) cannot be trusted: the wave marks itself and the repair launch recomputes its rows.
    auto v_phase = [&](const int t) __attribute__((always_inline)) {
      bool bad = false;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            // one operand fragment at a time (8 exp, 8 add, 4 cvt), fenced: left alone the scheduler issues all 64 exps
            // first and holds S and P in registers at the same time
            f32x16& sa = sacc[qb][kbk];
#pragma unroll
            for (int j = 0; j < 8; ++j) sa[8 * s + j] = fast_exp2(sa[8 * s + j]);
            a0 += sa[8 * s] + sa[8 * s + 4];
            a1 += sa[8 * s + 1] + sa[8 * s + 5];
            a2 += sa[8 * s + 2] + sa[8 * s + 6];
            a3 += sa[8 * s + 3] + sa[8 * s + 7];
            bf16x8 f = acc_to_frag(sa, s);
            {   // pin the fragment here: a value with one use in the M phase would otherwise be SUNK across the barrier
              typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
              u32x4 w = __builtin_bit_cast(u32x4, f);
              asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
              f = __builtin_bit_cast(bf16x8, w);
            }
            pf[qb][kbk][s] = f;
            __builtin_amdgcn_sched_barrier(0);
          }
        const float ls = (a0 + a1) + (a2 + a3);
        l_run[qb] += ls;
        bad |= !(ls < BIG);
      }
      poisoned |= bad;
    };
    // M phase of tile t: O += P(t) V(t);  S(t+1) = -m + K(t+1) Q^T
    auto m_phase = [&](const int t, const int slot) __attribute__((always_inline)) {
      const int vb = slot * 2 * TILE_BYTES + TILE_BYTES;
      PV_MFMAS(vb);
      // the seeds of S(t+1): 64 v_mov in the shadow of the MFMAs (the VALU is idle in this phase)
      SEED_S();
      if (t + 1 < nfast) {
        const int kb = (slot == NBUF - 1 ? 0 : slot + 1) * 2 * TILE_BYTES;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bf16x8 kf = ROW_FRAG(kb, kbk, s);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) sacc[qb][kbk] = mfma_bf16(kf, ROW_FRAG(qoff, qb, s), sacc[qb][kbk]);
          }
      }
    };

    // ring slot of tile t is t % 3, kept as a rotating scalar (a "% 3" of the loop counter makes the loop-strength
    // reduction pass turn every LDS address into its own induction variable)
    int slot = 1;
#define NEXT_SLOT(x_) ((x_) == NBUF - 1 ? 0 : (x_) + 1)
    if (role == 0) {
      for (int t = 1; t < nfast; ++t) {
        v_phase(t);                                      // phase 2t
        if (t >= 2 && t + 1 < nfast) STAGE_WRITE(NEXT_SLOT(slot));
        PHASE_BARRIER();
        if (t + 2 < nfast) STAGE_LOAD(t + 2);            // phase 2t+1
        m_phase(t, slot);
        PHASE_BARRIER();
        slot = NEXT_SLOT(slot);
      }
      PHASE_BARRIER();
    } else {
      PHASE_BARRIER();                                   // phase 2: idle
      for (int t = 1; t < nfast; ++t) {
        if (t + 2 < nfast) STAGE_LOAD(t + 2);            // phase 2t+1
        v_phase(t);
        PHASE_BARRIER();
        m_phase(t, slot);                                // phase 2t+2
        if (t + 2 < nfast) STAGE_WRITE(NEXT_SLOT(NEXT_SLOT(slot)));
        PHASE_BARRIER();
        slot = NEXT_SLOT(slot);
      }
    }
#undef NEXT_SLOT
  }
  PHASE_BARRIER();
#undef PACK_P
#undef ROW_FRAG
#undef SEED_S
#undef PV_MFMAS
#undef STAGE_LOAD
#undef STAGE_WRITE

  // ---- epilogue: normalise, stage O through LDS (per-wave 32 x 64 tile, 144-B rows), store whole rows.  The thread
  // index is laundered through an empty asm so that none of the addressing below is computed before the main loop
  // and carried through it in registers the loop needs.
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int wave_e = tid_e >> 6, lane_e = tid_e & 63, lr_e = lane_e & 31, lh_e = lane_e >> 5;
  const int q0_e = blockIdx.x * 512 + wave_e * 64;
  char* ob = smem + wave_e * (32 * 144);
  bf16_t* __restrict__ O = reinterpret_cast<bf16_t*>(p.out) + (int64_t)b * p.o_bs + h * 64;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = q0_e + qb * 32 + lr_e;
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = (l_tot > 0.f ? 1.f / l_tot : 0.f) * 1.f;   // normaliser uses the UNdropped sum
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 t4 = {(bf16_t)(oacc[qb][db][4 * g] * inv), (bf16_t)(oacc[qb][db][4 * g + 1] * inv),
                     (bf16_t)(oacc[qb][db][4 * g + 2] * inv), (bf16_t)(oacc[qb][db][4 * g + 3] * inv)};
        const int d = db * 32 + 8 * g + 4 * lh_e;
        *reinterpret_cast<bf16x4*>(ob + lr_e * 144 + d * 2) = t4;
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = i * 64 + lane_e, r = id >> 3, cc = id & 7;
      const int row = q0_e + qb * 32 + r;
      if (row < p.Lq)
        *reinterpret_cast<uint4*>(O + (int64_t)row * p.o_rs + cc * 8) = *reinterpret_cast<const uint4*>(ob + r * 144 + cc * 16);
    }
    if (lh_e == 0 && qrow < p.Lq) {
      const float m_use = (m_run[qb] == -INFINITY) ? 0.f : m_run[qb];
      // m is in log2 units; NaN = "recompute this row" for the repair launch
      p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = poisoned ? __builtin_nanf("") : m_use * 0.6931471805599453f + logf(l_tot);
    }
    if (qb == 0) __syncthreads();
  }
}

constexpr int PP_LDS_BYTES_PLACEHOLDER = 0;
}  // namespace

constexpr int PP_LDS_BYTES = NBUF * 2 * TILE_BYTES + 8 * 64 * 128;      // 48 KB ring + 64 KB Q = 112 KB of the CU's 160 KB

template <bool USE_IDX>
static hipError_t launch_pp(const AttnParams& p, hipStream_t st) {
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_pp_bf16_kernel<USE_IDX>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
  if (attr != hipSuccess) return attr;
  dim3 grid((p.Lq + 511) / 512, p.H, p.B), block(512);
  hipLaunchKernelGGL((attn_fwd_pp_bf16_kernel<USE_IDX>), grid, block, PP_LDS_BYTES, st, p);
  return hipGetLastError();
}

hipError_t launch_attn_fwd_pp_bf16(const AttnParams& p, hipStream_t st) {
  if (p.drop_thresh) return hipErrorInvalidValue;      // no dropout path in this experiment
  return p.kv_idx ? launch_pp<true>(p, st) : launch_pp<false>(p, st);
}
