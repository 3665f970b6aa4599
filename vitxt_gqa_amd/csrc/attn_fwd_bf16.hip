// bf16 flash-attention forward kernel for gfx950 (see attn_fwd.hip for the design notes and the
// reference call sites).  Templated on QB = number of 32-row query blocks per wave:
//   QB = 2: a wave owns 64 query rows (256 per workgroup); every K row fragment and V^T fragment read
//           from LDS feeds two MFMAs, halving LDS bytes and barriers per MFMA, and the two blocks'
//           softmax chains give the scheduler independent VALU work to put beside the MFMAs;
//   QB = 1: 32 rows per wave (128 per workgroup) for short sequences.
#include "attn_common.h"

namespace {

constexpr int BK = 64;    // keys per tile
constexpr int TILE_BYTES = BK * 128;

template <bool USE_IDX, int QB, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE_BYTES];   // [buf][K,V]
  constexpr int BQ = 128 * QB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * BQ + wave * (32 * QB);
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  const int ntiles = (nk + BK - 1) / BK;
  const bf16_t* __restrict__ Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* __restrict__ K = reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.kv_bs + h * 64;
  const bf16_t* __restrict__ V = reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.kv_bs + h * 64;
  const int32_t* __restrict__ idx = USE_IDX ? p.kv_idx + (int64_t)b * p.idx_cap : nullptr;

  // Q fragments: B operand of S^T = K Q^T; lane (q = lr, half lh) holds Q[q][16s + 8lh .. +7]
  bf16x8 qf[QB][4];
  int qdec[QB];             // decoder step of the lane's query row (negative: not a decoder row)
  uint32_t rk[QB], dsel[QB];   // dropout: row hash key and byte selector of the lane's query row
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qrow = q0 + qb * 32 + lr;
    const int qr = qrow < p.Lq ? qrow : p.Lq - 1;
    if (DROP) {
      rk[qb] = p.drop_rowkey[((int64_t)b * p.H + h) * ((p.Lq + 1) >> 1) + (qr >> 1)];
      dsel[qb] = (qr & 1) ? attn_drop_sel(2, 3) : attn_drop_sel(0, 1);
    }
    const bf16_t* qp = Q + (int64_t)qr * p.q_rs + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qb][s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    qdec[qb] = qrow - p.dec_q0;
  }

  // staging: thread -> (row sr / sr+32, 16-B chunk sc) of the K and V tiles.  Plain named registers and
  // unconditional (clamped) loads: arrays captured by a lambda or loads under a branch end up in scratch.
  const int sr = tid >> 3, sc = tid & 7;
  uint4 kr0, kr1, vr0, vr1;
#define STAGE_LOAD(t_)                                                                              \
  {                                                                                                 \
    int p0_ = (t_) * BK + sr, p1_ = p0_ + 32;                                                       \
    p0_ = p0_ < nk ? p0_ : nk - 1;                                                                  \
    p1_ = p1_ < nk ? p1_ : nk - 1;                                                                  \
    const int64_t r0_ = USE_IDX ? (int64_t)idx[p0_] : (int64_t)p0_;                                 \
    const int64_t r1_ = USE_IDX ? (int64_t)idx[p1_] : (int64_t)p1_;                                 \
    kr0 = *reinterpret_cast<const uint4*>(K + r0_ * p.kv_rs + sc * 8);                              \
    vr0 = *reinterpret_cast<const uint4*>(V + r0_ * p.kv_rs + sc * 8);                              \
    kr1 = *reinterpret_cast<const uint4*>(K + r1_ * p.kv_rs + sc * 8);                              \
    vr1 = *reinterpret_cast<const uint4*>(V + r1_ * p.kv_rs + sc * 8);                              \
  }
#define STAGE_WRITE(buf_)                                                                           \
  {                                                                                                 \
    char* kb_ = smem + (buf_) * 2 * TILE_BYTES;                                                     \
    *reinterpret_cast<uint4*>(kb_ + tile_off(sr, sc)) = kr0;                                        \
    *reinterpret_cast<uint4*>(kb_ + TILE_BYTES + tile_off(sr, sc)) = vr0;                           \
    *reinterpret_cast<uint4*>(kb_ + tile_off(sr + 32, sc)) = kr1;                                   \
    *reinterpret_cast<uint4*>(kb_ + TILE_BYTES + tile_off(sr + 32, sc)) = vr1;                      \
  }

  f32x16 oacc[QB][2];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[qb][0][i] = 0.f; oacc[qb][1][i] = 0.f; }
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
  }
  const float c = p.scale * LOG2E;

  if (ntiles > 0) {
    STAGE_LOAD(0);
    STAGE_WRITE(0);
  }
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    {
      const int tn = t + 1 < ntiles ? t + 1 : t;      // last iteration re-loads its own tile (harmless)
      STAGE_LOAD(tn);
    }
    const char* kb = smem + buf * 2 * TILE_BYTES;
    const char* vb = kb + TILE_BYTES;

    f32x16 sacc[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int i = 0; i < 16; ++i) { sacc[qb][0][i] = 0.f; sacc[qb][1][i] = 0.f; }
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = lds_row_frag(kb, kbk * 32 + lr, s, lh);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) sacc[qb][kbk] = mfma_bf16(kf, qf[qb][s], sacc[qb][kbk]);
      }

    // ---- online softmax over the 64 keys of this tile (this lane: 32 of them, partner lane^32 the rest)
    const bool edge = (t * BK + BK > n_prefix);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = -INFINITY;
      if (edge) {
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
      } else {
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[qb][kbk][r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qb], mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp2((m_run[qb] - m_use) * c);
      m_run[qb] = m_new;
      const float mc = m_use * c;
      float lsum = 0.f;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(sacc[qb][kbk][r] * c - mc);
          sacc[qb][kbk][r] = pv;
          lsum += pv;
        }
      l_run[qb] = l_run[qb] * alpha + lsum;
      // the running max rarely moves after the first tiles: skip the O rescale when no row of the wave changed
      if (!__all(alpha == 1.f)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[qb][0][i] *= alpha; oacc[qb][1][i] *= alpha; }
      }
    }

    // ---- O^T[d, q] += V^T[d, key] P^T[key, q]
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          pf[qb] = acc_to_frag(sacc[qb][kbk], s);
          if (DROP) {     // word i of the fragment = keys (2*kp2, 2*kp2 + 1) of this lane's query row
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 w = __builtin_bit_cast(u32x4, pf[qb]);
            const uint32_t th2 = p.drop_thresh | (p.drop_thresh << 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t kp2 = (uint32_t)(t * 32 + kbk * 16 + 8 * s + 4 * (i >> 1) + (i & 1) + 2 * lh);
              w[i] &= attn_drop_pair_mask(attn_drop_block(rk[qb], kp2), dsel[qb], th2);
            }
            pf[qb] = __builtin_bit_cast(bf16x8, w);
          }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 vf = lds_tr_frag(vb, kbk * 32 + 16 * s, db, lane);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) oacc[qb][db] = mfma_bf16(vf, pf[qb], oacc[qb][db]);
        }
      }

    STAGE_WRITE(buf ^ 1);
    __syncthreads();
  }
#undef STAGE_LOAD
#undef STAGE_WRITE

  // ---- epilogue: normalise, stage O through LDS (per-wave 32 x 64 tile, 144-B rows), store whole rows
  char* ob = smem + wave * (32 * 144);
  bf16_t* __restrict__ O = reinterpret_cast<bf16_t*>(p.out) + (int64_t)b * p.o_bs + h * 64;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
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
      p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = m_use * p.scale + logf(l_tot);
    }
    if (qb + 1 < QB) __syncthreads();
  }
}

}  // namespace

template <bool DROP>
static void launch_fwd(const AttnParams& p, hipStream_t st) {
  const bool wide = p.Lq > 256;       // 64 rows per wave once there is more than one workgroup of queries
  dim3 block(256);
  if (wide) {
    dim3 grid((p.Lq + 255) / 256, p.H, p.B);
    if (p.kv_idx) hipLaunchKernelGGL((attn_fwd_bf16_kernel<true, 2, DROP>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((attn_fwd_bf16_kernel<false, 2, DROP>), grid, block, 0, st, p);
  } else {
    dim3 grid((p.Lq + 127) / 128, p.H, p.B);
    if (p.kv_idx) hipLaunchKernelGGL((attn_fwd_bf16_kernel<true, 1, DROP>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((attn_fwd_bf16_kernel<false, 1, DROP>), grid, block, 0, st, p);
  }
}

void launch_attn_fwd_bf16(const AttnParams& p, hipStream_t st) {
  if (p.drop_thresh) launch_fwd<true>(p, st);
  else launch_fwd<false>(p, st);
}
