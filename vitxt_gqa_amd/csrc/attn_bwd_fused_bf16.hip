// Fused bf16 flash-attention backward for gfx950, head_dim 64: FIVE matrix products per (query, key) pair.
//
// The two-kernel form (attn_dkdv_bf16.hip + attn_dq_bf16_kernel) needs no cross-workgroup sum but computes S = Q K^T and
// dP = dO V^T twice: 7 products.  Here ONE key-stationary kernel computes S and dP once and also forms dQ += dS K; what it
// costs is a sum of dQ across the key blocks of a (sample, head), done with fp32 atomics.  Sizing that sum decides the
// geometry (cdna_hip_programming.md Guideline 12, MI355X_MICROARCH.md "Global float atomics": ~1.3 TB/s chip-wide):
// a workgroup of KB keys adds a [q-tile x 64] fp32 tile per query tile = 256 B per query row per 10*64*KB FLOPs, i.e.
// 2.5 * KB FLOP per atomic byte.  KB = 128 (the dK/dV kernel's block) caps the kernel at 0.42 PFLOP/s, below what the
// two-kernel form already reaches; KB = 384 puts the cap at 1.25 PFLOP/s.  So: 4 waves, ONE wave per SIMD with the whole
// 512-register file, 96 keys per wave:
//   registers  dK^T / dV^T of 96 keys (192 accumulators), V fragments (48), the query tile's A operands (64: Q, dO rows and
//              Q^T, dO^T transposed fragments, shared by the wave's three key blocks), row constants (32), S / dP (32)
//   LDS        K image of the workgroup's 384 keys (48 KB, pre-scaled by scale*log2e: B operand of S by row reads, B operand
//              of dQ by transposed reads), dS^T image [key][query] of the current query tile (48 KB), double-buffered Q / dO
//              tiles + row constants (33 KB)
// Per 64-row query tile:  phase A, per wave and 32-row sub-block: S, dP, P = exp2(S'), dS = P dP' (row constants seeded through
// the MFMA C operand), dV^T += dO^T P, dK^T += Q^T dS with P / dS straight from the accumulators (key on the lane), dS
// also stored transposed into the dS^T image (8 bytes per lane per 4 registers);  barrier;  phase B: wave w owns the
// 32 x 32 tile (query sub-block w >> 1, dim block w & 1) of dQ = dS K over ALL 384 keys (both operands by
// ds_read_b64_tr_b16 from the two images) and adds it to the fp32 dQ buffer: each accumulator register is two 128-byte
// row segments, the shape the atomics run at full rate for.
#include "attn_common.h"

namespace {

constexpr int FB_KB = 3;                          // 32-key blocks per wave
constexpr int FB_WKEYS = 32 * FB_KB;              // 96 keys per wave
constexpr int FB_KEYS = 4 * FB_WKEYS;             // 384 keys per workgroup
constexpr int FB_QROWS = 64;
constexpr int FB_TILE = FB_QROWS * 128;           // bytes of a 64-row bf16 tile
constexpr int FB_STAGE = 2 * FB_TILE + 2 * FB_QROWS * 4;   // Q | dO | -lse*log2e | -delta
constexpr int FB_KIMG = FB_KEYS * 128;
constexpr int FB_SMEM = 2 * FB_KIMG + 2 * FB_STAGE;

template <bool USE_IDX, bool TAIL>
__global__ __launch_bounds__(256, 1) void attn_bwd_fused_bf16_kernel(AttnParams p, float* __restrict__ dq32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const kimg = smem;                         // [384 keys][64 d] bf16, tile_off swizzle
  char* const dsimg = smem + FB_KIMG;              // [384 keys][64 q] bf16, same layout
  char* const stage = smem + 2 * FB_KIMG;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  int kblk, h, b;
  if (!attn_xcd_tile(TAIL ? 1 : p.kblocks, p.H, p.B, kblk, h, b)) return;         // workgroup-uniform (attn_common.h)
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  int kbw = TAIL ? p.kblocks : kblk;
  if (kbw * FB_KEYS >= nk) return;                 // uniform per workgroup
  const int32_t* __restrict__ idx = USE_IDX ? p.kv_idx + (int64_t)b * p.idx_cap : nullptr;
  const bf16_t* __restrict__ Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* __restrict__ DO = reinterpret_cast<const bf16_t*>(p.dout) + (int64_t)b * p.o_bs + h * 64;
  const char* __restrict__ Kg = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.kv_bs + h * 64);
  const char* __restrict__ Vg = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.kv_bs + h * 64);
  const float* __restrict__ LSE = p.lse + ((int64_t)b * p.H + h) * p.Lq;
  const float* __restrict__ DELTA = p.delta + ((int64_t)b * p.H + h) * p.Lq;
  float* __restrict__ DQ = dq32 + (int64_t)b * p.Lq * (p.H * 64) + h * 64;
  const float c = p.scale * LOG2E;
  const int nqt = (p.Lq + FB_QROWS - 1) / FB_QROWS;
  const int sr = tid >> 3, sc = tid & 7;

  do {   // key blocks of this workgroup (TAIL == false: exactly one, no loop is compiled)
    const int kp0 = kbw * FB_KEYS;
    const int nkeys_wg = (nk - kp0) < FB_KEYS ? (nk - kp0) : FB_KEYS;              // valid keys of this workgroup
    const int nks = (nkeys_wg + 15) >> 4;                                          // 16-key steps of the dQ product
    const bool edge_wg = (kp0 + FB_KEYS > n_prefix);                               // decoder keys or the end of the list inside
    // ---- K image of the workgroup's keys, pre-scaled by scale*log2e (one bf16 rounding per element, as the dK/dV kernel's
    // register fragments); rows past the list repeat its last key (their P is forced to 0 below)
#pragma unroll
    for (int i = 0; i < FB_KEYS / 32; ++i) {
      const int row = sr + 32 * i;
      int kp = kp0 + row;
      kp = kp < nk ? kp : nk - 1;
      const uint32_t grow = USE_IDX ? (uint32_t)idx[kp] : (uint32_t)kp;
      bf16x8 kv = *reinterpret_cast<const bf16x8*>(Kg + ((size_t)grow * (size_t)p.kv_rs + (size_t)sc * 8) * 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) kv[j] = (bf16_t)((float)kv[j] * c);
      *reinterpret_cast<bf16x8*>(kimg + tile_off(row, sc)) = kv;
    }
    // ---- this wave's keys: V fragments (B operands of dP), list positions, validity
    bf16x8 vf[FB_KB][4];
    int kdec[FB_KB];          // decoder step of this lane's key of block kb (negative: prefix key)
    bool kvalid[FB_KB];
    int64_t krow[FB_KB];
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb) {
      const int kpos = kp0 + wave * FB_WKEYS + kb * 32 + lr;
      kvalid[kb] = kpos < nk;
      const int kc = kvalid[kb] ? kpos : nk - 1;
      krow[kb] = USE_IDX ? (int64_t)idx[kc] : (int64_t)kc;
      kdec[kb] = kpos - n_prefix;
      const char* vp = Vg + (krow[kb] * p.kv_rs + 8 * lh) * 2;
#pragma unroll
      for (int s = 0; s < 4; ++s) vf[kb][s] = *reinterpret_cast<const bf16x8*>(vp + 32 * s);
    }
    f32x16 dkacc[FB_KB][2], dvacc[FB_KB][2];
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dkacc[kb][0][i] = 0.f; dkacc[kb][1][i] = 0.f; dvacc[kb][0][i] = 0.f; dvacc[kb][1][i] = 0.f; }

    // ---- staging of the Q / dO tiles (as attn_dkdv_bf16.hip: uniform 64-bit base + advancing 32-bit offsets, rows past Lq
    // clamp to the last row and get P = 0 through lse = -inf)
    uint4 q0r, q1r, d0r, d1r;
    float lreg, dreg;
    const char* __restrict__ Qb = reinterpret_cast<const char*>(Q);
    const char* __restrict__ DOb = reinterpret_cast<const char*>(DO);
    const uint32_t q_step = (uint32_t)(FB_QROWS * p.q_rs * 2), o_step = (uint32_t)(FB_QROWS * p.o_rs * 2);
    const uint32_t q_max = (uint32_t)((p.Lq - 1) * p.q_rs * 2) + (uint32_t)sc * 16u, o_max = (uint32_t)((p.Lq - 1) * p.o_rs * 2) + (uint32_t)sc * 16u;
    uint32_t qo0 = (uint32_t)(sr * p.q_rs * 2) + (uint32_t)sc * 16u, oo0 = (uint32_t)(sr * p.o_rs * 2) + (uint32_t)sc * 16u;
    uint32_t qo1 = qo0 + (uint32_t)(32 * p.q_rs * 2), oo1 = oo0 + (uint32_t)(32 * p.o_rs * 2);
    int ld_row = tid & 63;
#define FB_STAGE_LOAD()                                                                         \
  {                                                                                             \
    const uint32_t a0_ = qo0 < q_max ? qo0 : q_max, b0_ = oo0 < o_max ? oo0 : o_max;            \
    const uint32_t a1_ = qo1 < q_max ? qo1 : q_max, b1_ = oo1 < o_max ? oo1 : o_max;            \
    q0r = *reinterpret_cast<const uint4*>(Qb + a0_);                                            \
    d0r = *reinterpret_cast<const uint4*>(DOb + b0_);                                           \
    q1r = *reinterpret_cast<const uint4*>(Qb + a1_);                                            \
    d1r = *reinterpret_cast<const uint4*>(DOb + b1_);                                           \
    const int r2c_ = ld_row < p.Lq ? ld_row : p.Lq - 1;                                         \
    const float l_ = LSE[r2c_] * LOG2E, dl_ = DELTA[r2c_];                                      \
    lreg = ld_row < p.Lq ? -l_ : -INFINITY;                                                     \
    dreg = ld_row < p.Lq ? -dl_ : 0.f;                                                          \
    qo0 += q_step; oo0 += o_step; qo1 += q_step; oo1 += o_step;                                 \
    ld_row += FB_QROWS;                                                                         \
  }
#define FB_STAGE_WRITE(buf_)                                                                    \
  {                                                                                             \
    char* base_ = stage + (buf_) * FB_STAGE;                                                    \
    *reinterpret_cast<uint4*>(base_ + tile_off(sr, sc)) = q0r;                                  \
    *reinterpret_cast<uint4*>(base_ + FB_TILE + tile_off(sr, sc)) = d0r;                        \
    *reinterpret_cast<uint4*>(base_ + tile_off(sr + 32, sc)) = q1r;                             \
    *reinterpret_cast<uint4*>(base_ + FB_TILE + tile_off(sr + 32, sc)) = d1r;                   \
    if (tid < FB_QROWS) {                                                                       \
      reinterpret_cast<float*>(base_ + 2 * FB_TILE)[tid] = lreg;                                \
      reinterpret_cast<float*>(base_ + 2 * FB_TILE + FB_QROWS * 4)[tid] = dreg;                 \
    }                                                                                           \
  }
    FB_STAGE_LOAD();
    FB_STAGE_WRITE(0);
    __syncthreads();

    const int dq_qb = wave >> 1, dq_db = wave & 1;          // this wave's 32 x 32 tile of dQ
    for (int qt = 0; qt < nqt; ++qt) {
      const int buf = qt & 1;
      FB_STAGE_LOAD();                                      // next tile in sequence (past the end: clamped rows, harmless)
      const char* qb_ = stage + buf * FB_STAGE;
      const char* dob_ = qb_ + FB_TILE;
      const float* lse_s = reinterpret_cast<const float*>(qb_ + 2 * FB_TILE);
      const float* del_s = lse_s + FB_QROWS;
      // ================= phase A: S, dP, dS, dV^T, dK^T per (query sub-block, key block) =================
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        bf16x8 qf[4], dof[4], qT[2][2], doT[2][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          qf[s] = lds_row_frag(qb_, sb * 32 + lr, s, lh);
          dof[s] = lds_row_frag(dob_, sb * 32 + lr, s, lh);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            qT[s][db] = lds_tr_frag(qb_, sb * 32 + 16 * s, db, lane);
            doT[s][db] = lds_tr_frag(dob_, sb * 32 + 16 * s, db, lane);
          }
        f32x16 seed_s, seed_dp;          // row constants of this lane's accumulator rows: acc_row(r, lh) = 8g + 4lh + j
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + sb * 32 + 8 * g + 4 * lh);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + sb * 32 + 8 * g + 4 * lh);
#pragma unroll
          for (int j = 0; j < 4; ++j) { seed_s[4 * g + j] = l4[j]; seed_dp[4 * g + j] = d4[j]; }
        }
#pragma unroll
        for (int kb = 0; kb < FB_KB; ++kb) {
          const int keyrow0 = wave * FB_WKEYS + kb * 32;
          if (keyrow0 < nkeys_wg) {                          // wave-uniform: key blocks past the list are skipped
            f32x16 sacc = seed_s, dpacc = seed_dp;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              sacc = mfma_bf16(qf[s], lds_row_frag(kimg, keyrow0 + lr, s, lh), sacc);        // c*S[q, key] - LSE*log2e
              dpacc = mfma_bf16(dof[s], vf[kb][s], dpacc);                                    // dP[q, key] - delta
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = fast_exp2(sacc[r]);
            if (edge_wg) {          // decoder / validity rule behind a real uniform branch (see attn_dkdv_bf16_sweep.inc)
              asm volatile("" ::: "memory");
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int qdec = qt * FB_QROWS + sb * 32 + acc_row(r, lh) - p.dec_q0;
                const bool ok = kvalid[kb] && (kdec[kb] < 0 || qdec >= kdec[kb]);
                sacc[r] = ok ? sacc[r] : 0.f;
              }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dpacc[r] = sacc[r] * dpacc[r];
            char* dsrow = dsimg + 8 * lh;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              const bf16x8 pf = acc_to_frag(sacc, s), dsf = acc_to_frag(dpacc, s);
#pragma unroll
              for (int db = 0; db < 2; ++db) {
                dvacc[kb][db] = mfma_bf16(doT[s][db], pf, dvacc[kb][db]);       // dV^T[d, key] += dO^T[d, q] P[q, key]
                dkacc[kb][db] = mfma_bf16(qT[s][db], dsf, dkacc[kb][db]);       // dK^T[d, key] += Q^T[d, q] dS[q, key]
              }
              // dS^T image: this lane's key row, queries sb*32 + 16s + {0..3, 8..11} + 4lh: two 8-byte stores
              const uint4 w = __builtin_bit_cast(uint4, dsf);
              *reinterpret_cast<uint2*>(dsrow + tile_off(keyrow0 + lr, 4 * sb + 2 * s)) = make_uint2(w.x, w.y);
              *reinterpret_cast<uint2*>(dsrow + tile_off(keyrow0 + lr, 4 * sb + 2 * s + 1)) = make_uint2(w.z, w.w);
            }
          }
        }
      }
      __syncthreads();                                       // the dS^T image of this query tile is complete
      // ================= phase B: dQ[32 q, 32 d] of this wave over all keys of the workgroup =================
      {
        f32x16 dqacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) dqacc[i] = 0.f;
        for (int ks = 0; ks < nks; ++ks)
          dqacc = mfma_bf16(lds_tr_frag(dsimg, 16 * ks, dq_qb, lane), lds_tr_frag(kimg, 16 * ks, dq_db, lane), dqacc);
        // dS (K c) = c dS K;  dQ = scale dS K = acc * ln 2.  Register r = query row acc_row(r, lh), 32 consecutive dims per
        // half wave: two 128-byte segments per wave instruction
        const int q0 = qt * FB_QROWS + dq_qb * 32;
        float* dst = DQ + dq_db * 32 + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = q0 + acc_row(r, lh);
          if (q < p.Lq) unsafeAtomicAdd(dst + (int64_t)q * (p.H * 64), dqacc[r] * 0.6931471805599453f);
        }
      }
      FB_STAGE_WRITE(buf ^ 1);
      __syncthreads();                                       // next tile staged; every wave is done reading the dS^T image
    }
#undef FB_STAGE_LOAD
#undef FB_STAGE_WRITE

    // ---- dK / dV of this wave's keys
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb)
      if (kvalid[kb]) {
        bf16_t* dkp = reinterpret_cast<bf16_t*>(p.dk) + (int64_t)b * p.kv_bs + h * 64 + krow[kb] * p.kv_rs;
        bf16_t* dvp = reinterpret_cast<bf16_t*>(p.dv) + (int64_t)b * p.kv_bs + h * 64 + krow[kb] * p.kv_rs;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d = db * 32 + 8 * g + 4 * lh;
            bf16x4 k4 = {(bf16_t)(dkacc[kb][db][4 * g] * p.scale), (bf16_t)(dkacc[kb][db][4 * g + 1] * p.scale),
                         (bf16_t)(dkacc[kb][db][4 * g + 2] * p.scale), (bf16_t)(dkacc[kb][db][4 * g + 3] * p.scale)};
            bf16x4 v4 = {(bf16_t)dvacc[kb][db][4 * g], (bf16_t)dvacc[kb][db][4 * g + 1], (bf16_t)dvacc[kb][db][4 * g + 2],
                         (bf16_t)dvacc[kb][db][4 * g + 3]};
            *reinterpret_cast<bf16x4*>(dkp + d) = k4;
            *reinterpret_cast<bf16x4*>(dvp + d) = v4;
          }
      }
    if (TAIL) __syncthreads();                               // the next key block rewrites the K image
  } while (TAIL && (++kbw) * FB_KEYS < nk);
}

// delta[b, h, q] = sum_d dO * O (one wave per token row, as attn_delta_kernel), plus the housekeeping of the fused backward in
// the same pass over the rows: zero the row of the fp32 dQ accumulation buffer, and write the exact-zero dK / dV slices of
// prefix rows that no key-list entry points at (row_valid == 0).
__global__ __launch_bounds__(256) void attn_delta_prep_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, float* __restrict__ delta,
                                                              float* __restrict__ dq32, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv,
                                                              const uint8_t* __restrict__ row_valid, int valid_len, int B, int H, int Lq,
                                                              int64_t o_rs, int64_t o_bs, int64_t kv_rs, int64_t kv_bs) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * Lq) return;
  const int b = (int)(row / Lq), q = (int)(row % Lq);
  const bf16_t* op = o + (int64_t)b * o_bs + (int64_t)q * o_rs;
  const bf16_t* dp = dout + (int64_t)b * o_bs + (int64_t)q * o_rs;
  const int nchunk = H * 16;              // 4-element chunks per row
  float* zrow = dq32 + row * (int64_t)(H * 64);
  const bool fill = row_valid && q < valid_len && !row_valid[(int64_t)b * valid_len + q];
  for (int c0 = 0; c0 < nchunk; c0 += 64) {
    const int ci = c0 + lane;
    float s = 0.f;
    if (ci < nchunk) {
      const f32x4 a = Vec4<bf16_t>::load(op + ci * 4), d = Vec4<bf16_t>::load(dp + ci * 4);
      s = a[0] * d[0] + a[1] * d[1] + a[2] * d[2] + a[3] * d[3];
      *reinterpret_cast<f32x4*>(zrow + ci * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (fill) {
        const bf16x4 z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        *reinterpret_cast<bf16x4*>(dk + (int64_t)b * kv_bs + (int64_t)q * kv_rs + ci * 4) = z;
        *reinterpret_cast<bf16x4*>(dv + (int64_t)b * kv_bs + (int64_t)q * kv_rs + ci * 4) = z;
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if ((lane & 15) == 0 && ci < nchunk) delta[((int64_t)b * H + (ci >> 4)) * Lq + q] = s;
  }
}

// dq (bf16, strided rows inside the fused QKV gradient buffer) = bf16(dq32): 8 elements per thread
__global__ __launch_bounds__(256) void attn_dq_cast_kernel(const float* __restrict__ dq32, bf16_t* __restrict__ dq, int64_t rows_per_b, int width,
                                                           int64_t q_rs, int64_t q_bs, int64_t total8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total8) return;
  const int per_row = width >> 3;
  const int64_t row = i / per_row;
  const int c8 = (int)(i - row * per_row);
  const int64_t b = row / rows_per_b, q = row - b * rows_per_b;
  const f32x4 a = *reinterpret_cast<const f32x4*>(dq32 + row * width + c8 * 8);
  const f32x4 c = *reinterpret_cast<const f32x4*>(dq32 + row * width + c8 * 8 + 4);
  bf16x8 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)c[0], (bf16_t)c[1], (bf16_t)c[2], (bf16_t)c[3]};
  *reinterpret_cast<bf16x8*>(dq + b * q_bs + q * q_rs + c8 * 8) = o;
}

}  // namespace

// Fused backward (bf16, no attention dropout): delta + housekeeping, the 5-product kernel (+ its tail launch, see
// attn_dkdv_bf16.hip), the fp32 -> bf16 cast of dQ.
int launch_attn_bwd_fused_bf16(const AttnParams& p_in, int max_keys, float* dq32, hipStream_t st) {
  AttnParams p = p_in;
  // > 64 KB of LDS per workgroup needs the opt-in; set on every call (idempotent, per device, no state of ours is kept)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_bf16_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, FB_SMEM) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_bf16_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FB_SMEM) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_bf16_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, FB_SMEM) != hipSuccess) {
    t2s_set_error("attn_bwd_fused: cannot reserve %d bytes of LDS per workgroup", FB_SMEM);
    return 3;
  }
  const int64_t rows = (int64_t)p.B * p.Lq;
  hipLaunchKernelGGL(attn_delta_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)p.o, (const bf16_t*)p.dout, p.delta,
                     dq32, (bf16_t*)p.dk, (bf16_t*)p.dv, p.row_valid, p.valid_len, p.B, p.H, p.Lq, p.o_rs, p.o_bs, p.kv_rs, p.kv_bs);
  p.kblocks = (max_keys + FB_KEYS - 1) / FB_KEYS;
  dim3 grid(attn_xcd_grid(p.kblocks, p.H, p.B)), block(256), tail(attn_xcd_grid(1, p.H, p.B));
  if (p.kv_idx) {
    hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<true, false>), grid, block, FB_SMEM, st, p, dq32);
    if (p.kblocks * FB_KEYS < p.idx_cap) hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<true, true>), tail, block, FB_SMEM, st, p, dq32);
  } else {
    hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<false, false>), grid, block, FB_SMEM, st, p, dq32);
  }
  const int width = p.H * 64;
  const int64_t total8 = rows * (width / 8);
  hipLaunchKernelGGL(attn_dq_cast_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, st, dq32, (bf16_t*)p.dq, (int64_t)p.Lq, width,
                     p.q_rs, p.q_bs, total8);
  return 0;
}
