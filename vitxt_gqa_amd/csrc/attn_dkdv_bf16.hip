// dK / dV kernel of the bf16 flash-attention backward for gfx950 (design notes: attn_bwd.hip).
//
// Key-stationary: a workgroup = 4 waves = 128 keys of the compacted key list of one (batch, head);
// each wave keeps dK^T and dV^T of its 32 keys in accumulators (key on the MFMA lane) and holds its K / V
// rows as B-operand fragments in registers.  The workgroup sweeps the queries in tiles of 64 rows (two
// 32-row sub-blocks per barrier: half the barriers per MFMA, and two independent exp / dS chains for the
// scheduler to place beside the MFMAs) staged in LDS as Q and dO images plus LSE / delta vectors.
//   S = Q K^T, dP = dO V^T            (A = row reads of the LDS tiles, B = register fragments)
//   P = exp2(c S - LSE log2e), dS = P (dP - delta)
//   dV^T += dO^T P, dK^T += Q^T dS    (A = ds_read_b64_tr_b16 reads of the same tiles, B = P / dS accumulators)
#include "attn_common.h"

namespace {

constexpr int QROWS = 64;                         // query rows per iteration
constexpr int TILE = QROWS * 128;                 // bytes of a 64-row bf16 tile
constexpr int STAGE = 2 * TILE + 2 * QROWS * 4 + (QROWS / 2) * 4;   // Q | dO | -lse | -delta | dropout row keys

template <bool USE_IDX, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_dkdv_bf16_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  const int kp0 = blockIdx.x * 128;
  if (kp0 >= nk) return;                                   // uniform per workgroup
  const int kpos = kp0 + wave * 32 + lr;                   // this lane's key position (column)
  const bool kvalid = kpos < nk;
  const int32_t* __restrict__ idx = USE_IDX ? p.kv_idx + (int64_t)b * p.idx_cap : nullptr;
  const int kclamp = kvalid ? kpos : nk - 1;
  const int64_t krow = USE_IDX ? (int64_t)idx[kclamp] : (int64_t)kclamp;
  const bf16_t* __restrict__ Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* __restrict__ DO = reinterpret_cast<const bf16_t*>(p.dout) + (int64_t)b * p.o_bs + h * 64;
  const float* __restrict__ LSE = p.lse + ((int64_t)b * p.H + h) * p.Lq;
  const float* __restrict__ DELTA = p.delta + ((int64_t)b * p.H + h) * p.Lq;

  // K / V fragments of this wave's 32 keys: B operands, lane (key = lr, half lh) holds [key][16s+8lh..]
  bf16x8 kf[4], vf[4];
  {
    const bf16_t* kp = reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.kv_bs + h * 64 + krow * p.kv_rs + 8 * lh;
    const bf16_t* vp = reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.kv_bs + h * 64 + krow * p.kv_rs + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s);
      vf[s] = *reinterpret_cast<const bf16x8*>(vp + 16 * s);
    }
  }
  const int kdec = kpos - n_prefix;        // decoder step of this key (negative: prefix key)
  const float c = p.scale * LOG2E;
  // Fold the softmax scale into the K operand (one bf16 rounding per element, once per wave) and the per-row
  // constants into the accumulators' initial values: S'' = c*Q.K - LSE*log2e and dP' = dO.V - delta come straight
  // out of the MFMA chains, so P = exp2(S'') and dS = P * dP' need one v_exp and one v_mul per element.
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) kf[s][j] = (bf16_t)((float)kf[s][j] * c);
  const int nqt = (p.Lq + QROWS - 1) / QROWS;
  const bool edge = (kp0 + 128 > n_prefix);

  // staging: thread -> rows sr / sr+32, 16-B chunk sc of the Q and dO tiles; plain named registers and
  // unconditional clamped loads (keeps the staging out of scratch memory)
  const int sr = tid >> 3, sc = tid & 7;
  const int lrow = tid & 63;
  uint4 q0r, q1r, d0r, d1r;
  float lreg, dreg;
  uint32_t rkreg = 0;
  // dropout (attn_common.h): this lane's column key in both 16-bit halves; the row keys of the tile's 32 query pairs
  // are hashed by threads 0..31 while the tile is staged
  const uint32_t salt = DROP ? attn_drop_salt(p.drop_seed_lo, p.drop_seed_hi, (uint32_t)(b * p.H + h)) : 0u;
  const uint32_t ck2 = DROP ? attn_drop_colkey16(salt, kpos) * 0x10001u : 0u;
  const uint32_t th2 = attn_drop_thresh2s(p.drop_thresh);
#define STAGE_LOAD(qt_)                                                                         \
  {                                                                                             \
    const int r0_ = (qt_) * QROWS + sr, r1_ = r0_ + 32;                                         \
    const int c0_ = r0_ < p.Lq ? r0_ : p.Lq - 1, c1_ = r1_ < p.Lq ? r1_ : p.Lq - 1;             \
    q0r = *reinterpret_cast<const uint4*>(Q + (int64_t)c0_ * p.q_rs + sc * 8);                  \
    d0r = *reinterpret_cast<const uint4*>(DO + (int64_t)c0_ * p.o_rs + sc * 8);                 \
    q1r = *reinterpret_cast<const uint4*>(Q + (int64_t)c1_ * p.q_rs + sc * 8);                  \
    d1r = *reinterpret_cast<const uint4*>(DO + (int64_t)c1_ * p.o_rs + sc * 8);                 \
    if (r0_ >= p.Lq) d0r = make_uint4(0, 0, 0, 0);                                              \
    if (r1_ >= p.Lq) d1r = make_uint4(0, 0, 0, 0);                                              \
    const int r2_ = (qt_) * QROWS + lrow;                                                       \
    const int r2c_ = r2_ < p.Lq ? r2_ : p.Lq - 1;                                               \
    const float l_ = LSE[r2c_] * LOG2E, dl_ = DELTA[r2c_];                                      \
    lreg = r2_ < p.Lq ? -l_ : -INFINITY; /* -inf => P = exp2(-inf) = 0 for rows past Lq */       \
    dreg = r2_ < p.Lq ? -dl_ : 0.f;                                                             \
    if (DROP && tid < QROWS / 2) {                                                              \
      const int qa_ = (qt_) * QROWS + 2 * tid, qb_ = qa_ + 1;                                   \
      rkreg = attn_drop_rowkey16(salt, qa_ < p.Lq ? qa_ : p.Lq - 1) | (attn_drop_rowkey16(salt, qb_ < p.Lq ? qb_ : p.Lq - 1) << 16); \
    }                                                                                           \
  }
#define STAGE_WRITE(buf_)                                                                       \
  {                                                                                             \
    char* base_ = smem + (buf_) * STAGE;                                                        \
    *reinterpret_cast<uint4*>(base_ + tile_off(sr, sc)) = q0r;                                  \
    *reinterpret_cast<uint4*>(base_ + TILE + tile_off(sr, sc)) = d0r;                           \
    *reinterpret_cast<uint4*>(base_ + tile_off(sr + 32, sc)) = q1r;                             \
    *reinterpret_cast<uint4*>(base_ + TILE + tile_off(sr + 32, sc)) = d1r;                      \
    if (tid < QROWS) {                                                                          \
      reinterpret_cast<float*>(base_ + 2 * TILE)[tid] = lreg;                                   \
      reinterpret_cast<float*>(base_ + 2 * TILE + QROWS * 4)[tid] = dreg;                       \
    }                                                                                           \
    if (DROP && tid < QROWS / 2) reinterpret_cast<uint32_t*>(base_ + 2 * TILE + 2 * QROWS * 4)[tid] = rkreg; \
  }

  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dkacc[0][i] = 0.f; dkacc[1][i] = 0.f; dvacc[0][i] = 0.f; dvacc[1][i] = 0.f; }

  STAGE_LOAD(0);
  STAGE_WRITE(0);
  __syncthreads();
  for (int qt = 0; qt < nqt; ++qt) {
    const int buf = qt & 1;
    {
      const int qn = qt + 1 < nqt ? qt + 1 : qt;          // last iteration re-loads its own tile (harmless)
      STAGE_LOAD(qn);
    }
    const char* qb = smem + buf * STAGE;
    const char* dob = qb + TILE;
    const float* lse_s = reinterpret_cast<const float*>(qb + 2 * TILE);
    const float* del_s = lse_s + QROWS;
    const uint32_t* rk_s = reinterpret_cast<const uint32_t*>(del_s + QROWS);

    f32x16 sacc[2], dpacc[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      // initial accumulators = row constants (rows of this lane's registers: acc_row(r, lh) = 8g + 4lh + j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + sb * 32 + 8 * g + 4 * lh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + sb * 32 + 8 * g + 4 * lh);
#pragma unroll
        for (int j = 0; j < 4; ++j) { sacc[sb][4 * g + j] = l4[j]; dpacc[sb][4 * g + j] = DROP ? 0.f : d4[j]; }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sacc[sb] = mfma_bf16(lds_row_frag(qb, sb * 32 + lr, s, lh), kf[s], sacc[sb]);        // c*S[q, key] - LSE*log2e
        dpacc[sb] = mfma_bf16(lds_row_frag(dob, sb * 32 + lr, s, lh), vf[s], dpacc[sb]);     // dP[q, key] - delta
      }
    }
    uint32_t pfw[2][8], dsw[2][8];       // dropout variant: packed bf16 operand words of P (dropped) and dS
    if (!DROP) {
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pv = fast_exp2(sacc[sb][r]);
          if (edge) {
            const int qdec = qt * QROWS + sb * 32 + acc_row(r, lh) - p.dec_q0;
            const bool ok = kvalid && (kdec < 0 || qdec >= kdec);
            pv = ok ? pv : 0.f;
          }
          sacc[sb][r] = pv;
          dpacc[sb][r] = pv * dpacc[sb][r];
        }
    } else {
      // dA = dD * M / (1 - p);  dS = P * (dA - delta) with the UNdropped P;  dV uses the dropped P.  Registers (r, r+1),
      // r even, are the query rows (qi, qi + 1) of this lane's key = one bf16x2 operand word: one packed mask word per
      // pair clears the dropped halves of the packed P, and M/(1-p) enters dS as a float that is 1/(1-p) or 0
      // (bit-and of the constant with the sign-extended mask half): exp, fma, mul per score plus the shared mask work.
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      const uint32_t inv_bits = __builtin_bit_cast(uint32_t, p.drop_inv);
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int qi = sb * 32 + acc_row(r, lh);
          float pv0 = fast_exp2(sacc[sb][r]), pv1 = fast_exp2(sacc[sb][r + 1]);
          if (edge) {
            const int qdec = qt * QROWS + qi - p.dec_q0;
            pv0 = (kvalid && (kdec < 0 || qdec >= kdec)) ? pv0 : 0.f;
            pv1 = (kvalid && (kdec < 0 || qdec + 1 >= kdec)) ? pv1 : 0.f;
          }
          const uint32_t m = attn_drop_pair_dropped(rk_s[qi >> 1], ck2, th2);
          const float g0 = __builtin_bit_cast(float, inv_bits & ~attn_drop_lo32(m));
          const float g1 = __builtin_bit_cast(float, inv_bits & ~attn_drop_hi32(m));
          const f32x2 nd = *reinterpret_cast<const f32x2*>(del_s + qi);
          const bf16x2_t pw = {(__bf16)pv0, (__bf16)pv1};
          const bf16x2_t dw = {(__bf16)(pv0 * __builtin_fmaf(dpacc[sb][r], g0, nd[0])), (__bf16)(pv1 * __builtin_fmaf(dpacc[sb][r + 1], g1, nd[1]))};
          pfw[sb][r >> 1] = attn_drop_apply(__builtin_bit_cast(uint32_t, pw), m);
          dsw[sb][r >> 1] = __builtin_bit_cast(uint32_t, dw);
        }
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const bf16x8 pf = DROP ? __builtin_bit_cast(bf16x8, u32x4{pfw[sb][4 * s], pfw[sb][4 * s + 1], pfw[sb][4 * s + 2], pfw[sb][4 * s + 3]})
                               : acc_to_frag(sacc[sb], s);
        const bf16x8 dsf = DROP ? __builtin_bit_cast(bf16x8, u32x4{dsw[sb][4 * s], dsw[sb][4 * s + 1], dsw[sb][4 * s + 2], dsw[sb][4 * s + 3]})
                                : acc_to_frag(dpacc[sb], s);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dvacc[db] = mfma_bf16(lds_tr_frag(dob, sb * 32 + 16 * s, db, lane), pf, dvacc[db]);   // dV^T[d,key] += dO^T[d,q] P[q,key]
          dkacc[db] = mfma_bf16(lds_tr_frag(qb, sb * 32 + 16 * s, db, lane), dsf, dkacc[db]);   // dK^T[d,key] += Q^T[d,q] dS[q,key]
        }
      }
    STAGE_WRITE(buf ^ 1);
    __syncthreads();
  }
#undef STAGE_LOAD
#undef STAGE_WRITE

  if (kvalid) {
    bf16_t* dkp = reinterpret_cast<bf16_t*>(p.dk) + (int64_t)b * p.kv_bs + h * 64 + krow * p.kv_rs;
    bf16_t* dvp = reinterpret_cast<bf16_t*>(p.dv) + (int64_t)b * p.kv_bs + h * 64 + krow * p.kv_rs;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * lh;
        bf16x4 k4 = {(bf16_t)(dkacc[db][4 * g] * p.scale), (bf16_t)(dkacc[db][4 * g + 1] * p.scale),
                     (bf16_t)(dkacc[db][4 * g + 2] * p.scale), (bf16_t)(dkacc[db][4 * g + 3] * p.scale)};
        const float vs_ = DROP ? p.drop_inv : 1.f;
        bf16x4 v4 = {(bf16_t)(dvacc[db][4 * g] * vs_), (bf16_t)(dvacc[db][4 * g + 1] * vs_), (bf16_t)(dvacc[db][4 * g + 2] * vs_),
                     (bf16_t)(dvacc[db][4 * g + 3] * vs_)};
        *reinterpret_cast<bf16x4*>(dkp + d) = k4;
        *reinterpret_cast<bf16x4*>(dvp + d) = v4;
      }
  }
}

}  // namespace

void launch_attn_dkdv_bf16(const AttnParams& p, int max_keys, hipStream_t st) {
  dim3 grid((max_keys + 127) / 128, p.H, p.B), block(256);
  if (p.drop_thresh) {
    if (p.kv_idx) hipLaunchKernelGGL((attn_dkdv_bf16_kernel<true, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((attn_dkdv_bf16_kernel<false, true>), grid, block, 0, st, p);
  } else {
    if (p.kv_idx) hipLaunchKernelGGL((attn_dkdv_bf16_kernel<true, false>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((attn_dkdv_bf16_kernel<false, false>), grid, block, 0, st, p);
  }
}
