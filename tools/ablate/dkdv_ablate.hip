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

namespace {

constexpr int QROWS = 64;                        // query rows per iteration
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
  const bool has_dec = p.n_dec > 0 && (kp0 + 128 > n_prefix);    // this workgroup holds decoder keys

  // staging: thread -> rows sr / sr+32, 16-B chunk sc of the Q and dO tiles; plain named registers and
  // unconditional clamped loads (keeps the staging out of scratch memory)
  const int sr = tid >> 3, sc = tid & 7;
  const int lrow = tid & 63;
  uint4 q0r, q1r, d0r, d1r;
  float lreg, dreg;
  uint32_t rkreg = 0;
  const int Lq2 = (p.Lq + 1) >> 1;
  const uint32_t* __restrict__ RK = DROP ? p.drop_rowkey + ((int64_t)b * p.H + h) * Lq2 : nullptr;
  const uint32_t kp2 = (uint32_t)kpos >> 1;
  const uint32_t ksel = attn_drop_sel(kpos & 1, 2 + (kpos & 1));      // bytes (q even, q odd) of this lane's key
  const uint32_t th2 = p.drop_thresh | (p.drop_thresh << 16);
// STAGE_LOAD issues the raw global loads only (no use of the loaded values, so they can stay in flight across the
// whole body); STAGE_WRITE applies the row-past-Lq fix-ups and writes the tile.
#define STAGE_LOAD(qt_)                                                                         \
  {                                                                                             \
    const int r0_ = (qt_) * QROWS + sr, r1_ = r0_ + 32;                                         \
    const int c0_ = r0_ < p.Lq ? r0_ : p.Lq - 1, c1_ = r1_ < p.Lq ? r1_ : p.Lq - 1;             \
    q0r = *reinterpret_cast<const uint4*>(Q + (int64_t)c0_ * p.q_rs + sc * 8);                  \
    d0r = *reinterpret_cast<const uint4*>(DO + (int64_t)c0_ * p.o_rs + sc * 8);                 \
    q1r = *reinterpret_cast<const uint4*>(Q + (int64_t)c1_ * p.q_rs + sc * 8);                  \
    d1r = *reinterpret_cast<const uint4*>(DO + (int64_t)c1_ * p.o_rs + sc * 8);                 \
    const int r2_ = (qt_) * QROWS + lrow;                                                       \
    const int r2c_ = r2_ < p.Lq ? r2_ : p.Lq - 1;                                               \
    lreg = LSE[r2c_];                                                                           \
    dreg = DELTA[r2c_];                                                                         \
    if (DROP) {                                                                                 \
      const int q2_ = (qt_) * (QROWS / 2) + (tid & 31);                                         \
      rkreg = RK[q2_ < Lq2 ? q2_ : Lq2 - 1];                                                    \
    }                                                                                           \
  }
#define STAGE_WRITE(buf_, qt_)                                                                  \
  {                                                                                             \
    char* base_ = smem + (buf_) * STAGE;                                                        \
    const int r0_ = (qt_) * QROWS + sr, r1_ = r0_ + 32, r2_ = (qt_) * QROWS + lrow;             \
    if (r0_ >= p.Lq) d0r = make_uint4(0, 0, 0, 0);                                              \
    if (r1_ >= p.Lq) d1r = make_uint4(0, 0, 0, 0);                                              \
    *reinterpret_cast<uint4*>(base_ + tile_off(sr, sc)) = q0r;                                  \
    *reinterpret_cast<uint4*>(base_ + TILE + tile_off(sr, sc)) = d0r;                           \
    *reinterpret_cast<uint4*>(base_ + tile_off(sr + 32, sc)) = q1r;                             \
    *reinterpret_cast<uint4*>(base_ + TILE + tile_off(sr + 32, sc)) = d1r;                      \
    /* every wave writes the same 64 row constants (no branch); -inf => P = exp2(-inf) = 0 for rows past Lq */ \
    reinterpret_cast<float*>(base_ + 2 * TILE)[lrow] = r2_ < p.Lq ? -lreg * LOG2E : -INFINITY;  \
    reinterpret_cast<float*>(base_ + 2 * TILE + QROWS * 4)[lrow] = r2_ < p.Lq ? -dreg : 0.f;    \
    if (DROP && tid < QROWS / 2) reinterpret_cast<uint32_t*>(base_ + 2 * TILE + 2 * QROWS * 4)[tid] = rkreg; \
  }

  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dkacc[0][i] = 0.f; dkacc[1][i] = 0.f; dvacc[0][i] = 0.f; dvacc[1][i] = 0.f; }

  STAGE_LOAD(0);
  STAGE_WRITE(0, 0);
  __syncthreads();
  // The query sweep exists in two compiled forms selected by ONE workgroup-uniform branch: only the workgroup that
  // holds the decoder keys needs the causal rule.  (Left inside the loop, the rule is if-converted into 32 compares,
  // 40 selects and ~100 scalar ops per iteration for every workgroup.)  Lanes whose key lies past the end of the list
  // need no masking at all: with the key on the lane, their garbage stays in their own dK/dV columns, which are never
  // stored.
  auto sweep = [&](auto masked_tag) {
  constexpr bool MASKED = decltype(masked_tag)::value;
  for (int qt = 0; qt < nqt; ++qt) {
    const int buf = qt & 1;
    const int qn = qt + 1 < nqt ? qt + 1 : qt;            // last iteration re-loads its own tile (harmless)
    if (!(ABL & 1)) { STAGE_LOAD(qn); }
    if (!(ABL & 64)) __builtin_amdgcn_sched_barrier(0);                    // keep the next tile's global loads at the top of the body
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
        sacc[sb] = mfma_bf16((ABL & 4) ? vf[s] : lds_row_frag(qb, sb * 32 + lr, s, lh), kf[s], sacc[sb]);        // c*S[q, key] - LSE*log2e
        dpacc[sb] = mfma_bf16((ABL & 4) ? kf[s] : lds_row_frag(dob, sb * 32 + lr, s, lh), vf[s], dpacc[sb]);     // dP[q, key] - delta
      }
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pv = (ABL & 2) ? sacc[sb][r] : fast_exp2(sacc[sb][r]);
        if (MASKED) {     // decoder key j is visible to query row r iff r - dec_q0 >= j
          const int qdec = qt * QROWS + sb * 32 + acc_row(r, lh) - p.dec_q0;
          pv = (kdec < 0 || qdec >= kdec) ? pv : 0.f;
        }
        sacc[sb][r] = pv;
        if (DROP) {      // dA = dD * M / (1 - p);  dS = P * (dA - delta) with the UNdropped P
          const int qi = sb * 32 + acc_row(r, lh);
          const uint32_t x = attn_drop_block(rk_s[qi >> 1], kp2);
          const bool keep = ((x >> (8 * ((qi & 1) * 2 + (kpos & 1)))) & 0xFFu) >= p.drop_thresh;
          dpacc[sb][r] = pv * ((keep ? dpacc[sb][r] * p.drop_inv : 0.f) + del_s[qi]);
        } else {
          dpacc[sb][r] = pv * dpacc[sb][r];
        }
      }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf = acc_to_frag(sacc[sb], s);
        const bf16x8 dsf = acc_to_frag(dpacc[sb], s);
        if (DROP) {      // dV uses the dropped probabilities: word i = query rows (2*q2, 2*q2 + 1) of this lane's key
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          u32x4 w = __builtin_bit_cast(u32x4, pf);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int q2i = sb * 16 + (i & 1) + 4 * (2 * s + (i >> 1)) + 2 * lh;
            w[i] &= attn_drop_pair_mask(attn_drop_block(rk_s[q2i], kp2), ksel, th2);
          }
          pf = __builtin_bit_cast(bf16x8, w);
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dvacc[db] = mfma_bf16((ABL & 8) ? kf[db] : lds_tr_frag(dob, sb * 32 + 16 * s, db, lane), pf, dvacc[db]);   // dV^T[d,key] += dO^T[d,q] P[q,key]
          dkacc[db] = mfma_bf16((ABL & 8) ? vf[db] : lds_tr_frag(qb, sb * 32 + 16 * s, db, lane), dsf, dkacc[db]);   // dK^T[d,key] += Q^T[d,q] dS[q,key]
        }
      }
    if (!(ABL & 16)) { STAGE_WRITE(buf ^ 1, qn); }
    if (!(ABL & 32)) __syncthreads();
  }
  };
  if (has_dec) sweep(std::true_type{});
  else sweep(std::false_type{});
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

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int B = 8, H = 12, L = 10132, ND = 12;
  const size_t nqkv = (size_t)B * L * 3 * 768, no = (size_t)B * L * 768, nl = (size_t)B * H * L;
  std::vector<uint16_t> h(nqkv);
  srand(1);
  for (size_t i = 0; i < nqkv; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; uint32_t u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  void *qkv, *dout, *dqkv; float *lse, *delta;
  CK(hipMalloc(&qkv, nqkv * 2)); CK(hipMalloc(&dqkv, nqkv * 2)); CK(hipMalloc(&dout, no * 2));
  CK(hipMalloc(&lse, nl * 4)); CK(hipMalloc(&delta, nl * 4));
  CK(hipMemcpy(qkv, h.data(), nqkv * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dout, h.data(), no * 2, hipMemcpyHostToDevice));
  std::vector<float> l(nl, 12.0f);
  CK(hipMemcpy(lse, l.data(), nl * 4, hipMemcpyHostToDevice));
  CK(hipMemset(delta, 0, nl * 4));
  AttnParams p{};
  p.q = qkv; p.k = (char*)qkv + 768 * 2; p.v = (char*)qkv + 2 * 768 * 2; p.dout = dout;
  p.dq = dqkv; p.dk = (char*)dqkv + 768 * 2; p.dv = (char*)dqkv + 2 * 768 * 2;
  p.lse = lse; p.delta = delta; p.kv_idx = nullptr; p.kv_cnt = nullptr;
  p.B = B; p.H = H; p.Lq = L; p.idx_cap = L; p.n_dec = ND; p.dec_q0 = L - ND;
  p.q_rs = 3 * 768; p.q_bs = (int64_t)L * 3 * 768; p.kv_rs = 3 * 768; p.kv_bs = p.q_bs; p.o_rs = 768; p.o_bs = (int64_t)L * 768;
  p.scale = 0.125f; p.drop_thresh = 0; p.drop_inv = 1.f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch_attn_dkdv_bf16(p, L, 0); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 5; ++i) launch_attn_dkdv_bf16(p, L, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double fl = 5.0 * 2.0 * B * H * (double)L * L * 64;   // 5 GEMM-shaped products
  printf("ABL=%d dkdv %.3f ms  %.1f TF/s\n", ABL, ms / 5, fl / (ms / 5 * 1e-3) / 1e12);
  return 0;
}
