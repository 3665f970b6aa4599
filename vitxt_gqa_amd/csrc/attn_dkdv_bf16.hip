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
#include <type_traits>

#include "attn_common.h"

#define T2S_DKDV_SWEEP "attn_dkdv_bf16_sweep.inc"

namespace {

constexpr int QROWS = 64;                         // query rows per iteration
constexpr int TILE = QROWS * 128;                 // bytes of a 64-row bf16 tile
constexpr int STAGE = 2 * TILE + 2 * QROWS * 4 + (QROWS / 2) * 4;   // Q | dO | -lse | -delta | dropout row keys

// NW = waves per workgroup (4 or 8): a workgroup owns 32 * NW keys and its waves share every staged Q / dO tile, so with
// NW = 8 each thread stages half as much per MFMA (one row chunk of Q and of dO per tile instead of two) and the tiles are
// fetched from L2 half as often; occupancy is the same two waves per SIMD (one 8-wave workgroup per CU instead of two
// 4-wave ones).
#ifndef T2S_DKDV_NW
#define T2S_DKDV_NW 4
#endif
template <bool USE_IDX, bool DROP, int NW, bool TAIL>
__global__ __launch_bounds__(64 * NW, 2) void attn_dkdv_bf16_kernel(AttnParams p) {
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  constexpr int KEYS = 32 * NW;                    // keys per workgroup
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  int kblk, h, b;
  if (!attn_xcd_tile(TAIL ? 1 : p.kblocks, p.H, p.B, kblk, h, b)) return;         // workgroup-uniform (attn_common.h)
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  // The main grid covers p.kblocks key blocks per (sample, head): a host-side STATIC bound on the key count (max_keys).  A
  // second, one-workgroup-per-(sample, head) TAIL launch of the same kernel walks whatever key blocks lie beyond it, so a
  // bound that does not hold (injected grounding masks, a dataset whose temporal ids do not follow the P-slots-per-frame
  // layout) costs time, never gradients; where the bound holds the tail workgroups exit at once.  (As a loop inside the main
  // kernel the same guarantee spilled 6 VGPRs in the dropout variants.)
  int kb = TAIL ? p.kblocks : kblk;
  if (kb * KEYS >= nk) return;                             // uniform per workgroup
  do {
  const int kp0 = kb * KEYS;
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
  const bool edge_wg = (kp0 + KEYS > n_prefix);      // this workgroup holds decoder keys or the end of the list

  // staging: thread -> rows sr / sr+32, 16-B chunk sc of the Q and dO tiles; plain named registers and unconditional
  // clamped loads (keeps the staging out of scratch memory).  Addresses are a workgroup-uniform 64-bit base plus a 32-bit
  // per-thread byte offset that ADVANCES by a uniform step per tile (the first form of this macro recomputed
  // row * stride in 64 bits for every load: 8 v_mul_lo_u32 + 4 v_mad_u64_u32 + 6 v_lshl_add_u64 per tile, all multi-pass
  // instructions, ~13 % of the kernel); rows past Lq clamp to the last row by a v_min on the offset and get P = 0 through
  // lse = -inf, so their (finite) Q / dO values never reach a sum.  A sample's rows span < 4 GB (checked by the host).
  const int sr = tid >> 3, sc = tid & 7;
  const int lrow = tid & 63;
  uint4 q0r, q1r, d0r, d1r;
  float lreg, dreg;
  uint32_t rkreg = 0;
  const char* __restrict__ Qb = reinterpret_cast<const char*>(Q);
  const char* __restrict__ DOb = reinterpret_cast<const char*>(DO);
  const uint32_t q_step = (uint32_t)(QROWS * p.q_rs * 2), o_step = (uint32_t)(QROWS * p.o_rs * 2);       // bytes per tile
  const uint32_t q_max = (uint32_t)((p.Lq - 1) * p.q_rs * 2) + (uint32_t)sc * 16u, o_max = (uint32_t)((p.Lq - 1) * p.o_rs * 2) + (uint32_t)sc * 16u;
  uint32_t qo0 = (uint32_t)(sr * p.q_rs * 2) + (uint32_t)sc * 16u, oo0 = (uint32_t)(sr * p.o_rs * 2) + (uint32_t)sc * 16u;
  uint32_t qo1 = qo0 + (uint32_t)(32 * p.q_rs * 2), oo1 = oo0 + (uint32_t)(32 * p.o_rs * 2);
  int ld_row = lrow;            // query row whose lse / delta this thread fetches for the tile being loaded
  int ld_row0 = 0;              // first query row of that tile (uniform)
  // dropout (attn_common.h): this lane's column key in both 16-bit halves; the row keys of the tile's 32 query pairs
  // are hashed by threads 0..31 while the tile is staged
  const uint32_t salt = DROP ? attn_drop_salt(p.drop_seed_lo, p.drop_seed_hi, (uint32_t)(b * p.H + h)) : 0u;
  uint32_t ck2 = 0u;                                      // this lane's column key of the current 256-row window (set in the sweep)
  const uint32_t th2 = attn_drop_thresh2k(p.drop_thresh);      // (keep words: attn_drop_pair_kept)
  // loads the NEXT tile in sequence (tile 0 first; past the last tile: clamped copies of the last row, harmless)
#define STAGE_LOAD(qt_unused_)                                                                  \
  {                                                                                             \
    const uint32_t a0_ = qo0 < q_max ? qo0 : q_max, b0_ = oo0 < o_max ? oo0 : o_max;            \
    q0r = *reinterpret_cast<const uint4*>(Qb + a0_);                                            \
    d0r = *reinterpret_cast<const uint4*>(DOb + b0_);                                           \
    if (NW == 4) {      /* 256 threads: a second row per thread */                              \
      const uint32_t a1_ = qo1 < q_max ? qo1 : q_max, b1_ = oo1 < o_max ? oo1 : o_max;          \
      q1r = *reinterpret_cast<const uint4*>(Qb + a1_);                                          \
      d1r = *reinterpret_cast<const uint4*>(DOb + b1_);                                         \
    }                                                                                           \
    const int r2c_ = ld_row < p.Lq ? ld_row : p.Lq - 1;                                         \
    lreg = LSE[r2c_]; dreg = DELTA[r2c_];       /* raw: arithmetic on them HERE would make the wave wait for the loads here */  \
    if (DROP && tid < QROWS / 2) {                                                              \
      const int qa_ = ld_row0 + 2 * tid, qb_ = qa_ + 1;                                         \
      rkreg = attn_drop_rowkey16(salt, qa_ < p.Lq ? qa_ : p.Lq - 1, kp0 / ATTN_DROP_KWIN) |                                            \
              (attn_drop_rowkey16(salt, qb_ < p.Lq ? qb_ : p.Lq - 1, kp0 / ATTN_DROP_KWIN) << 16);       /* (the block's keys lie in one window) */ \
    }                                                                                           \
    qo0 += q_step; oo0 += o_step; qo1 += q_step; oo1 += o_step;                                 \
    ld_row += QROWS; ld_row0 += QROWS;                                                          \
  }
#define STAGE_WRITE(buf_)                                                                       \
  {                                                                                             \
    char* base_ = smem + (buf_) * STAGE;                                                        \
    *reinterpret_cast<uint4*>(base_ + tile_off(sr, sc)) = q0r;                                  \
    *reinterpret_cast<uint4*>(base_ + TILE + tile_off(sr, sc)) = d0r;                           \
    if (NW == 4) {                                                                              \
      *reinterpret_cast<uint4*>(base_ + tile_off(sr + 32, sc)) = q1r;                           \
      *reinterpret_cast<uint4*>(base_ + TILE + tile_off(sr + 32, sc)) = d1r;                    \
    }                                                                                           \
    if (tid < QROWS) {           /* (ld_row was advanced past the staged tile by the load); -inf => P = exp2(-inf) = 0 for rows past Lq */ \
      const bool in_ = ld_row - QROWS < p.Lq;                                                   \
      reinterpret_cast<float*>(base_ + 2 * TILE)[tid] = in_ ? -(lreg * LOG2E) : -INFINITY;      \
      reinterpret_cast<float*>(base_ + 2 * TILE + QROWS * 4)[tid] = in_ ? -dreg : 0.f;          \
    }                                                                                           \
    if (DROP && tid < QROWS / 2) reinterpret_cast<uint32_t*>(base_ + 2 * TILE + 2 * QROWS * 4)[tid] = rkreg; \
  }

  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dkacc[0][i] = 0.f; dkacc[1][i] = 0.f; dvacc[0][i] = 0.f; dvacc[1][i] = 0.f; }

  STAGE_LOAD(0);
  STAGE_WRITE(0);
  __syncthreads();
  // With dropout the sweep is compiled in two forms selected by ONE workgroup-uniform branch: the decoder / validity rule
  // if-converts into ~170 VALU instructions per iteration, paid by every workgroup although only the last key block of
  // a sample needs it (535 -> 375 VALU per 32 MFMAs).  Without dropout the whole-loop split measured 5 % slower; there the
  // rule sits behind a uniform branch INSIDE the one loop (see the sweep).
  if (!DROP) {
    const bool edge = edge_wg;
#include T2S_DKDV_SWEEP
  } else if (!edge_wg) {
    constexpr bool edge = false;
#include T2S_DKDV_SWEEP
  } else {
    constexpr bool edge = true;
#include T2S_DKDV_SWEEP
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
  } while (TAIL && (++kb) * KEYS < nk);   // key blocks of this workgroup (TAIL == false: exactly one, no loop is compiled)
}

}  // namespace

void launch_attn_dkdv_bf16(const AttnParams& p_in, int max_keys, hipStream_t st) {
  constexpr int NW = T2S_DKDV_NW;
  AttnParams p = p_in;
  p.kblocks = (max_keys + 32 * NW - 1) / (32 * NW);
  dim3 grid(attn_xcd_grid(p.kblocks, p.H, p.B)), block(64 * NW);      // XCD-aware 1-D grid (attn_common.h)
  dim3 tail(attn_xcd_grid(1, p.H, p.B));
  const bool need_tail = p.kv_idx != nullptr && p.kblocks * 32 * NW < p.idx_cap;      // a dense list's length is known exactly
  if (p.drop_thresh) {
    if (p.kv_idx) {
      hipLaunchKernelGGL((attn_dkdv_bf16_kernel<true, true, NW, false>), grid, block, 0, st, p);
      if (need_tail) hipLaunchKernelGGL((attn_dkdv_bf16_kernel<true, true, NW, true>), tail, block, 0, st, p);
    } else hipLaunchKernelGGL((attn_dkdv_bf16_kernel<false, true, NW, false>), grid, block, 0, st, p);
  } else {
    if (p.kv_idx) {
      hipLaunchKernelGGL((attn_dkdv_bf16_kernel<true, false, NW, false>), grid, block, 0, st, p);
      if (need_tail) hipLaunchKernelGGL((attn_dkdv_bf16_kernel<true, false, NW, true>), tail, block, 0, st, p);
    } else hipLaunchKernelGGL((attn_dkdv_bf16_kernel<false, false, NW, false>), grid, block, 0, st, p);
  }
}
