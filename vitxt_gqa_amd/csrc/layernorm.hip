// residual-add + LayerNorm over rows of 768 (BertSelfOutput / BertOutput / BertLayerNorm of the
// third-party BERT block; also pythia/models/t2s.py:87-88 (obj_feat_layer_norm), :116-117
// (ocr_feat/ocr_bbox_layer_norm), :685-687 (PrevPredEmbeddings)).  Biased variance, eps inside sqrt.
//
// Mixed precision: the branch input x (a GEMM output) and the residual stream (res, y, z) have
// separate storage types.  In the bf16 compute mode the residual stream stays fp32 (x: bf16,
// res/y/z: fp32) and a second bf16 copy of y (y_lo) is emitted for the next GEMM, so rounding to
// bf16 happens only at GEMM/attention operands, never on the running hidden state.
//
// HBM-bound: one wavefront per row, 12 elements per lane as three 4-element vectors (8-B loads in
// bf16, 16-B in fp32, lane-contiguous => 512 B / 1 KiB coalesced per instruction), statistics by
// 64-lane shuffle reduction (two-pass mean / centred variance in registers: the row is read once).
// Algorithmic bytes per row: fwd = 768 * (read x, res + write y, z [+ y_lo]); bwd = 768 * (dy + z + dz).
#include "common.h"

namespace {

constexpr int H = T2S_HIDDEN;           // 768 = 64 lanes * 3 vectors * 4
constexpr int ROWS_PER_BLOCK = 4;       // 4 waves per workgroup
constexpr int BWD_MAX_PARTS = 2048;

// Hidden-state dropout of BertSelfOutput / BertOutput (dropout(dense(.)) before the residual add), fused here:
// keep(row, col) is a stateless hash of (seed, row*768 + col), so backward regenerates the same mask from the seed.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
struct DropCfg {
  uint32_t seed_lo, seed_hi, thresh;     // drop iff (hash >> 8) < thresh, thresh = p * 2^24
  float scale;                           // 1 / (1 - p); thresh == 0 disables dropout
};
__device__ __forceinline__ float drop_keep_scale(const DropCfg& d, uint32_t idx) {
  const uint32_t h = hash32(hash32(idx + d.seed_lo) ^ d.seed_hi);
  return (h >> 8) < d.thresh ? 0.f : d.scale;
}

template <typename TX, typename TS>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const TX* __restrict__ x, const TS* __restrict__ res,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                TS* __restrict__ y, bf16_t* __restrict__ y_lo, TS* z_out,
                                                                float* __restrict__ stats, int64_t rows, float eps, DropCfg drop,
                                                                const float* __restrict__ res_stats, const float* __restrict__ res_gamma,
                                                                const float* __restrict__ res_beta) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[3];
  float s = 0.f;
  // normalised residual: `res` holds the previous block's pre-LayerNorm sum and is normalised on the fly with that
  // block's statistics and affine (the same expression that block's own output used), so the fp32 stream value never
  // makes a round trip through HBM
  float rmean = 0.f, rrstd = 1.f;
  if (res_stats) { rmean = res_stats[row * 2]; rrstd = res_stats[row * 2 + 1]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = (i * 64 + lane) * 4;
    v[i] = Vec4<TX>::load(x + row * H + e);
    if (drop.thresh) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] *= drop_keep_scale(drop, (uint32_t)(row * H + e + j));
    }
    if (res) {
      f32x4 r = Vec4<TS>::load(res + row * H + e);
      if (res_stats) {
        const f32x4 rg = *reinterpret_cast<const f32x4*>(res_gamma + e);
        const f32x4 rb = *reinterpret_cast<const f32x4*>(res_beta + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (r[j] - rmean) * rrstd * rg[j] + rb[j];
      }
      v[i] += r;
    }
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  if (z_out) {
#pragma unroll
    for (int i = 0; i < 3; ++i) Vec4<TS>::store(z_out + row * H + (i * 64 + lane) * 4, v[i]);
  }
  const float mean = wave_sum(s) * (1.f / H);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = v[i][j] - mean;
      sq += d * d;
    }
  const float rstd = 1.f / sqrtf(wave_sum(sq) * (1.f / H) + eps);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = (i * 64 + lane) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + e);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + e);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
    if (y) Vec4<TS>::store(y + row * H + e, o);
    if (y_lo) Vec4<bf16_t>::store(y_lo + row * H + e, o);
  }
  if (stats && lane == 0) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
}

// Backward: dz = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)).  Each workgroup walks rows with a
// grid stride and keeps per-lane partial sums of dgamma/dbeta (12 columns per lane) in registers; the
// 4 waves are combined through LDS and written as one [768] partial row per workgroup.
template <typename TDY, typename TZ, typename TDZ>
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const TDY* __restrict__ dy, const TZ* __restrict__ z,
                                                                const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                TDZ* __restrict__ dz, TDZ* __restrict__ dzx,
                                                                float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                                float* __restrict__ dbias_part, int64_t rows, DropCfg drop) {
  __shared__ float red[3][4][H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 g[3], dg[3], db[3], dbi[3];      // dbi: column sums of the branch-input gradient (= bias gradient of the dense layer before)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g[i] = *reinterpret_cast<const f32x4*>(gamma + (i * 64 + lane) * 4);
    dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    dbi[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave; row < rows; row += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    f32x4 d[3], xh[3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = (i * 64 + lane) * 4;
      d[i] = Vec4<TDY>::load(dy + row * H + e);
      const f32x4 zz = Vec4<TZ>::load(z + row * H + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[i][j] = (zz[j] - mean) * rstd;
        dg[i][j] += d[i][j] * xh[i][j];
        db[i][j] += d[i][j];
        const float t = d[i][j] * g[i][j];
        d[i][j] = t;
        s1 += t;
        s2 += t * xh[i][j];
      }
    }
    s1 = wave_sum(s1) * (1.f / H);
    s2 = wave_sum(s2) * (1.f / H);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = rstd * (d[i][j] - s1 - xh[i][j] * s2);
      Vec4<TDZ>::store(dz + row * H + (i * 64 + lane) * 4, o);
      if (dzx) {             // gradient of the dropped branch input: dz * keep / (1 - p)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] *= drop_keep_scale(drop, (uint32_t)(row * H + (i * 64 + lane) * 4 + j));
        Vec4<TDZ>::store(dzx + row * H + (i * 64 + lane) * 4, o);
      }
      if (dbias_part) dbi[i] += o;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[0][wave][(i * 64 + lane) * 4 + j] = dg[i][j];
      red[1][wave][(i * 64 + lane) * 4 + j] = db[i][j];
      red[2][wave][(i * 64 + lane) * 4 + j] = dbi[i][j];
    }
  __syncthreads();
  for (int cix = threadIdx.x; cix < H; cix += 256) {
    dgamma_part[(int64_t)blockIdx.x * H + cix] = red[0][0][cix] + red[0][1][cix] + red[0][2][cix] + red[0][3][cix];
    dbeta_part[(int64_t)blockIdx.x * H + cix] = red[1][0][cix] + red[1][1][cix] + red[1][2][cix] + red[1][3][cix];
    if (dbias_part) dbias_part[(int64_t)blockIdx.x * H + cix] = red[2][0][cix] + red[2][1][cix] + red[2][2][cix] + red[2][3][cix];
  }
}

// ---- tail of the OCR-token encoding (T2S._forward_ocr_encoding, pythia/models/t2s.py:221-258):
//   out = dropout( LN_feat(a) + LN_bbox(bbox W_b^T + b_b) )
// with a [rows, 768] the output of linear_ocr_feat_to_mmt_in (a library GEMM) and bbox [rows, 4].  As framework ops this is two
// LayerNorm calls (each keeping an fp32 copy of its input for backward), a K = 4 "GEMM" that writes 2 GB of fp32, an add and a
// dropout: ~26 GB of HBM traffic at B=64 forward and about as much backward.  Here the 768-wide box projection is recomputed per
// row from its 4 inputs (48 weights per lane in registers), both rows are normalised in registers, and one fp32 row is written:
// 1 (bf16 a) + 2 GB.  Backward reads the output gradient and a once, rebuilds both normalised rows, writes the gradient of a and
// accumulates the six parameter gradients (LN affines, box weight and bias) in registers across a grid-stride loop.
struct TailStats { float mean_a, rstd_a, mean_b, rstd_b; };

template <typename TA>
__global__ __launch_bounds__(256) void ocr_tail_fwd_kernel(const TA* __restrict__ a, const float* __restrict__ bbox, const float* __restrict__ wb,
                                                           const float* __restrict__ bb, const float* __restrict__ ga, const float* __restrict__ ba,
                                                           const float* __restrict__ gb, const float* __restrict__ bb2, float* __restrict__ out,
                                                           float* __restrict__ stats, int64_t rows, float eps, DropCfg drop) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // this lane's 12 columns: c = (i * 64 + lane) * 4 + j.  W_b is [768, 4] row-major: the 4 weights of a column are one float4
  f32x4 w[3][4], bias[3], gA[3], bA[3], gB[3], bB[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = (i * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[i][j] = *reinterpret_cast<const f32x4*>(wb + (e + j) * 4);
    bias[i] = *reinterpret_cast<const f32x4*>(bb + e);
    gA[i] = *reinterpret_cast<const f32x4*>(ga + e);
    bA[i] = *reinterpret_cast<const f32x4*>(ba + e);
    gB[i] = *reinterpret_cast<const f32x4*>(gb + e);
    bB[i] = *reinterpret_cast<const f32x4*>(bb2 + e);
  }
  for (int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave; row < rows; row += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
    const f32x4 x4 = *reinterpret_cast<const f32x4*>(bbox + row * 4);
    f32x4 va[3], vb[3];
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      va[i] = Vec4<TA>::load(a + row * H + (i * 64 + lane) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // nn.Linear: sum_k x[k] W[c, k] + b[c], accumulated in k order like the library's K = 4 dot product
        vb[i][j] = fmaf(x4[3], w[i][j][3], fmaf(x4[2], w[i][j][2], fmaf(x4[1], w[i][j][1], x4[0] * w[i][j][0]))) + bias[i][j];
        sa += va[i][j];
        sb += vb[i][j];
      }
    }
    const float ma = wave_sum(sa) * (1.f / H), mb = wave_sum(sb) * (1.f / H);
    float qa = 0.f, qb = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float da = va[i][j] - ma, db = vb[i][j] - mb;
        qa += da * da;
        qb += db * db;
      }
    const float ra = 1.f / sqrtf(wave_sum(qa) * (1.f / H) + eps), rb = 1.f / sqrtf(wave_sum(qb) * (1.f / H) + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = (i * 64 + lane) * 4;
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = ((va[i][j] - ma) * ra * gA[i][j] + bA[i][j]) + ((vb[i][j] - mb) * rb * gB[i][j] + bB[i][j]);
        if (drop.thresh) o[j] *= drop_keep_scale(drop, (uint32_t)(row * H + e + j));
      }
      *reinterpret_cast<f32x4*>(out + row * H + e) = o;
    }
    if (lane == 0) *reinterpret_cast<f32x4*>(stats + row * 4) = f32x4{ma, ra, mb, rb};
  }
}

// partial sums per workgroup, [n_part, 9 * 768] fp32: dgamma_a | dbeta_a | dgamma_b | dbeta_b | dbias_box | dW_box (4 x 768, k-major)
constexpr int TAIL_PARTS = 9;
template <typename TA>
__global__ __launch_bounds__(256, 2) void ocr_tail_bwd_kernel(const float* __restrict__ gout, const TA* __restrict__ a, const float* __restrict__ bbox,
                                                              const float* __restrict__ wb, const float* __restrict__ bb, const float* __restrict__ ga,
                                                              const float* __restrict__ gb, const float* __restrict__ stats, TA* __restrict__ da_out,
                                                              float* __restrict__ part, int64_t rows, DropCfg drop) {
  // the per-column constants (box weight 4 + bias + two LayerNorm gains = 7 floats per column) live in LDS and are re-read per row:
  // with them in registers beside the 108 partial sums the kernel needs 256 VGPRs (one wave per SIMD, 3.7 TB/s); from LDS it keeps
  // two waves per SIMD in flight
  __shared__ float cst[7][H];
  __shared__ float red[4][H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < H; c += 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k) cst[k][c] = wb[c * 4 + k];          // k-major: the lane's 4 consecutive columns are one float4 per k
    cst[4][c] = bb[c];
    cst[5][c] = ga[c];
    cst[6][c] = gb[c];
  }
  __syncthreads();
  f32x4 acc[TAIL_PARTS][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int q = 0; q < TAIL_PARTS; ++q) acc[q][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave; row < rows; row += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
    const f32x4 x4 = *reinterpret_cast<const f32x4*>(bbox + row * 4);
    const f32x4 st = *reinterpret_cast<const f32x4*>(stats + row * 4);
    // two sweeps over the lane's 12 columns: the first forms the row sums, the second rebuilds the normalised values from the
    // registers that hold g and a (the constants come from LDS again) - keeping xa / xb / ta / tb of the whole row alive costs 48 VGPRs
    f32x4 g[3], av[3];
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
    asm volatile("" ::: "memory");        // the constants are row-invariant: without this the compiler hoists their 84 LDS loads into registers
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = (i * 64 + lane) * 4;
      g[i] = *reinterpret_cast<const f32x4*>(gout + row * H + e);
      av[i] = Vec4<TA>::load(a + row * H + e);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(&cst[0][e]), w1 = *reinterpret_cast<const f32x4*>(&cst[1][e]);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(&cst[2][e]), w3 = *reinterpret_cast<const f32x4*>(&cst[3][e]);
      const f32x4 bi = *reinterpret_cast<const f32x4*>(&cst[4][e]);
      const f32x4 gA = *reinterpret_cast<const f32x4*>(&cst[5][e]), gB = *reinterpret_cast<const f32x4*>(&cst[6][e]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (drop.thresh) g[i][j] *= drop_keep_scale(drop, (uint32_t)(row * H + e + j));
        const float bv = fmaf(x4[3], w3[j], fmaf(x4[2], w2[j], fmaf(x4[1], w1[j], x4[0] * w0[j]))) + bi[j];
        const float xa = (av[i][j] - st[0]) * st[1], xb = (bv - st[2]) * st[3];
        acc[0][i][j] += g[i][j] * xa;
        acc[1][i][j] += g[i][j];
        acc[2][i][j] += g[i][j] * xb;
        acc[3][i][j] += g[i][j];
        const float ta = g[i][j] * gA[j], tb = g[i][j] * gB[j];
        s1a += ta;
        s2a += ta * xa;
        s1b += tb;
        s2b += tb * xb;
      }
    }
    s1a = wave_sum(s1a) * (1.f / H);
    s2a = wave_sum(s2a) * (1.f / H);
    s1b = wave_sum(s1b) * (1.f / H);
    s2b = wave_sum(s2b) * (1.f / H);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = (i * 64 + lane) * 4;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(&cst[0][e]), w1 = *reinterpret_cast<const f32x4*>(&cst[1][e]);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(&cst[2][e]), w3 = *reinterpret_cast<const f32x4*>(&cst[3][e]);
      const f32x4 bi = *reinterpret_cast<const f32x4*>(&cst[4][e]);
      const f32x4 gA = *reinterpret_cast<const f32x4*>(&cst[5][e]), gB = *reinterpret_cast<const f32x4*>(&cst[6][e]);
      f32x4 oa;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bv = fmaf(x4[3], w3[j], fmaf(x4[2], w2[j], fmaf(x4[1], w1[j], x4[0] * w0[j]))) + bi[j];
        const float xa = (av[i][j] - st[0]) * st[1], xb = (bv - st[2]) * st[3];
        oa[j] = st[1] * (g[i][j] * gA[j] - s1a - xa * s2a);
        const float ob = st[3] * (g[i][j] * gB[j] - s1b - xb * s2b);      // gradient of the box projection's output
        acc[4][i][j] += ob;
        acc[5][i][j] += ob * x4[0];
        acc[6][i][j] += ob * x4[1];
        acc[7][i][j] += ob * x4[2];
        acc[8][i][j] += ob * x4[3];
      }
      Vec4<TA>::store(da_out + row * H + e, oa);
    }
  }
  // 4 waves -> one partial row per workgroup and quantity (through LDS, one quantity at a time)
#pragma unroll
  for (int q = 0; q < TAIL_PARTS; ++q) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave][(i * 64 + lane) * 4 + j] = acc[q][i][j];
    __syncthreads();
    for (int cix = threadIdx.x; cix < H; cix += 256)
      part[((int64_t)blockIdx.x * TAIL_PARTS + q) * H + cix] = red[0][cix] + red[1][cix] + red[2][cix] + red[3][cix];
  }
}

// Pre-LayerNorm block of the ViT frame-feature producer (tools/video_feat/obtain_vit_feat.py:37-53 runs transformers' ViTLayer:
// x = x + attn(LN(x)); x = x + mlp(LN(x))): the residual update and the NEXT LayerNorm in one pass over the rows of the fp32 stream.
//   h <- h + branch + bias   (branch: the bf16 GEMM output of the block just finished, bias: that dense layer's fp32 bias; both optional)
//   y  = LN(h) * gamma + beta  in bf16 (the next GEMM's operand) or fp32 (the final LayerNorm)
// Row width W <= 256 * NV, a multiple of 4 (ViT-L: 1024 = 4 vectors per lane; lanes behind W idle), one wavefront per row, the row
// held in registers.
template <int NV, typename TB, typename TY>
__global__ __launch_bounds__(256) void wide_add_layernorm_fwd_kernel(float* __restrict__ h, const TB* __restrict__ branch,
                                                                     const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, TY* __restrict__ y, int64_t rows,
                                                                     int64_t h_stride, int W, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = (i * 64 + lane) * 4;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (e < W) {
      v[i] = *reinterpret_cast<const f32x4*>(h + row * h_stride + e);
      if (branch) {
        v[i] += Vec4<TB>::load(branch + row * W + e);
        if (bias) v[i] += *reinterpret_cast<const f32x4*>(bias + e);
        *reinterpret_cast<f32x4*>(h + row * h_stride + e) = v[i];
      }
    }
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  const float inv_w = 1.f / (float)W;
  const float mean = wave_sum(s) * inv_w;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if ((i * 64 + lane) * 4 < W) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[i][j] - mean;
        sq += d * d;
      }
    }
  const float rstd = 1.f / sqrtf(wave_sum(sq) * inv_w + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = (i * 64 + lane) * 4;
    if (e < W) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + e);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + e);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      Vec4<TY>::store(y + row * W + e, o);
    }
  }
}

int bwd_parts(int64_t rows) {
  int64_t n = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return (int)(n < BWD_MAX_PARTS ? (n < 1 ? 1 : n) : BWD_MAX_PARTS);
}

bool is_dt(int d) { return d == T2S_F32 || d == T2S_BF16; }

DropCfg make_drop(float p, uint64_t seed) {
  DropCfg d;
  d.seed_lo = (uint32_t)seed;
  d.seed_hi = (uint32_t)(seed >> 32);
  d.thresh = p > 0.f ? (uint32_t)(p * 16777216.0f) : 0u;
  d.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  return d;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ out, int64_t n, DropCfg drop) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = drop_keep_scale(drop, (uint32_t)i) != 0.f ? 1 : 0;
}

}  // namespace

static int add_layernorm_fwd_impl(const void* x, const void* res, const float* res_stats, const float* res_gamma, const float* res_beta,
                                  const float* gamma, const float* beta, void* y, void* y_lo,
                                  void* z_out, float* stats, int64_t rows, float eps, int x_dtype, int stream_dtype,
                                  float drop_p, uint64_t drop_seed, t2s_stream_t stream) {
  T2S_CHECK_ARG(x && gamma && beta && (y || y_lo), "add_layernorm_fwd: null pointer");
  T2S_CHECK_ARG(rows > 0 && rows < ((int64_t)1 << 33), "add_layernorm_fwd: bad rows %lld", (long long)rows);
  T2S_CHECK_ARG(is_dt(x_dtype) && is_dt(stream_dtype), "add_layernorm_fwd: bad dtype %d/%d", x_dtype, stream_dtype);
  T2S_CHECK_ARG(!(x_dtype == T2S_F32 && stream_dtype == T2S_BF16), "add_layernorm_fwd: fp32 branch with bf16 stream is not supported");
  T2S_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "add_layernorm_fwd: dropout probability %f outside [0, 1)", drop_p);
  T2S_CHECK_ARG(drop_p == 0.f || rows * H < ((int64_t)1 << 32), "add_layernorm_fwd: dropout index space exceeds 32 bits");
  const DropCfg drop = make_drop(drop_p, drop_seed);
  dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == T2S_BF16 && stream_dtype == T2S_BF16)
    hipLaunchKernelGGL((add_layernorm_fwd_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)res, gamma, beta,
                       (bf16_t*)y, (bf16_t*)y_lo, (bf16_t*)z_out, stats, rows, eps, drop, res_stats, res_gamma, res_beta);
  else if (x_dtype == T2S_BF16)
    hipLaunchKernelGGL((add_layernorm_fwd_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)x, (const float*)res, gamma, beta,
                       (float*)y, (bf16_t*)y_lo, (float*)z_out, stats, rows, eps, drop, res_stats, res_gamma, res_beta);
  else
    hipLaunchKernelGGL((add_layernorm_fwd_kernel<float, float>), grid, block, 0, st, (const float*)x, (const float*)res, gamma, beta,
                       (float*)y, (bf16_t*)y_lo, (float*)z_out, stats, rows, eps, drop, res_stats, res_gamma, res_beta);
  T2S_CHECK_LAUNCH("add_layernorm_fwd");
  return 0;
}

extern "C" int t2s_add_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta, void* y, void* y_lo,
                                     void* z_out, float* stats, int64_t rows, float eps, int x_dtype, int stream_dtype,
                                     float drop_p, uint64_t drop_seed, t2s_stream_t stream) {
  return add_layernorm_fwd_impl(x, res, nullptr, nullptr, nullptr, gamma, beta, y, y_lo, z_out, stats, rows, eps, x_dtype, stream_dtype,
                                drop_p, drop_seed, stream);
}

extern "C" int t2s_add_layernorm_fwd_nres(const void* x, const void* res_z, const float* res_stats, const float* res_gamma,
                                          const float* res_beta, const float* gamma, const float* beta, void* y, void* y_lo,
                                          void* z_out, float* stats, int64_t rows, float eps, int x_dtype, int stream_dtype,
                                          float drop_p, uint64_t drop_seed, t2s_stream_t stream) {
  T2S_CHECK_ARG(res_z && res_stats && res_gamma && res_beta, "add_layernorm_fwd_nres: null pointer in the normalised residual");
  return add_layernorm_fwd_impl(x, res_z, res_stats, res_gamma, res_beta, gamma, beta, y, y_lo, z_out, stats, rows, eps, x_dtype,
                                stream_dtype, drop_p, drop_seed, stream);
}

extern "C" int t2s_layernorm_bwd_parts(int64_t rows) { return bwd_parts(rows); }

static int add_layernorm_bwd_impl(const void* dy, const void* z, const float* stats, const float* gamma, void* dz, void* dzx,
                                  float* dgamma_part, float* dbeta_part, float* dbias_part, int64_t rows, int dy_dtype, int z_dtype,
                                  int dz_dtype, float drop_p, uint64_t drop_seed, t2s_stream_t stream) {
  T2S_CHECK_ARG(dy && z && stats && gamma && dz && dgamma_part && dbeta_part, "add_layernorm_bwd: null pointer");
  T2S_CHECK_ARG(rows > 0, "add_layernorm_bwd: bad rows");
  T2S_CHECK_ARG(is_dt(dy_dtype) && is_dt(z_dtype) && is_dt(dz_dtype), "add_layernorm_bwd: bad dtype");
  T2S_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (dzx != nullptr) == (drop_p > 0.f), "add_layernorm_bwd: dzx must be given iff drop_p > 0");
  const DropCfg drop = make_drop(drop_p, drop_seed);
  dim3 grid(bwd_parts(rows)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int combo = dy_dtype * 4 + z_dtype * 2 + dz_dtype;
#define LN_BWD(TDY, TZ, TDZ)                                                                                                   \
  hipLaunchKernelGGL((add_layernorm_bwd_kernel<TDY, TZ, TDZ>), grid, block, 0, st, (const TDY*)dy, (const TZ*)z, stats, gamma, \
                     (TDZ*)dz, (TDZ*)dzx, dgamma_part, dbeta_part, dbias_part, rows, drop)
  switch (combo) {
    case 0: LN_BWD(float, float, float); break;
    case 1: LN_BWD(float, float, bf16_t); break;
    case 5: LN_BWD(bf16_t, float, bf16_t); break;
    case 7: LN_BWD(bf16_t, bf16_t, bf16_t); break;
    default:
      t2s_set_error("add_layernorm_bwd: unsupported dtype combination dy=%d z=%d dz=%d", dy_dtype, z_dtype, dz_dtype);
      return 1;
  }
#undef LN_BWD
  T2S_CHECK_LAUNCH("add_layernorm_bwd");
  return 0;
}

extern "C" int t2s_add_layernorm_bwd(const void* dy, const void* z, const float* stats, const float* gamma, void* dz, void* dzx,
                                     float* dgamma_part, float* dbeta_part, int64_t rows, int dy_dtype, int z_dtype, int dz_dtype,
                                     float drop_p, uint64_t drop_seed, t2s_stream_t stream) {
  return add_layernorm_bwd_impl(dy, z, stats, gamma, dz, dzx, dgamma_part, dbeta_part, nullptr, rows, dy_dtype, z_dtype, dz_dtype, drop_p,
                                drop_seed, stream);
}

extern "C" int t2s_add_layernorm_bwd_bias(const void* dy, const void* z, const float* stats, const float* gamma, void* dz, void* dzx,
                                          float* dgamma_part, float* dbeta_part, float* dbias_part, int64_t rows, int dy_dtype,
                                          int z_dtype, int dz_dtype, float drop_p, uint64_t drop_seed, t2s_stream_t stream) {
  T2S_CHECK_ARG(dbias_part, "add_layernorm_bwd_bias: null dbias_part");
  return add_layernorm_bwd_impl(dy, z, stats, gamma, dz, dzx, dgamma_part, dbeta_part, dbias_part, rows, dy_dtype, z_dtype, dz_dtype,
                                drop_p, drop_seed, stream);
}

extern "C" int t2s_dropout_mask(uint8_t* out, int64_t n, float drop_p, uint64_t drop_seed, t2s_stream_t stream) {
  T2S_CHECK_ARG(out && n > 0 && n < ((int64_t)1 << 32), "dropout_mask: bad arguments");
  T2S_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "dropout_mask: bad probability");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, make_drop(drop_p, drop_seed));
  T2S_CHECK_LAUNCH("dropout_mask");
  return 0;
}


extern "C" int t2s_ocr_tail_parts(int64_t rows) { return bwd_parts(rows); }

extern "C" int t2s_ocr_tail_fwd(const void* a, int a_dtype, const float* bbox, const float* w_box, const float* b_box, const float* gamma_a,
                                const float* beta_a, const float* gamma_b, const float* beta_b, float* out, float* stats, int64_t rows, float eps,
                                float drop_p, uint64_t drop_seed, t2s_stream_t stream) {
  T2S_CHECK_ARG(a && bbox && w_box && b_box && gamma_a && beta_a && gamma_b && beta_b && out && stats, "ocr_tail_fwd: null pointer");
  T2S_CHECK_ARG(rows > 0 && is_dt(a_dtype), "ocr_tail_fwd: bad shape / dtype");
  T2S_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rows * H < ((int64_t)1 << 32)), "ocr_tail_fwd: bad dropout");
  const DropCfg drop = make_drop(drop_p, drop_seed);
  int64_t nb = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (nb > 8192) nb = 8192;
  dim3 grid((unsigned)nb), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (a_dtype == T2S_BF16)
    hipLaunchKernelGGL(ocr_tail_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)a, bbox, w_box, b_box, gamma_a, beta_a, gamma_b, beta_b, out, stats,
                       rows, eps, drop);
  else
    hipLaunchKernelGGL(ocr_tail_fwd_kernel<float>, grid, block, 0, st, (const float*)a, bbox, w_box, b_box, gamma_a, beta_a, gamma_b, beta_b, out, stats,
                       rows, eps, drop);
  T2S_CHECK_LAUNCH("ocr_tail_fwd");
  return 0;
}

extern "C" int t2s_ocr_tail_bwd(const float* g_out, const void* a, int a_dtype, const float* bbox, const float* w_box, const float* b_box,
                                const float* gamma_a, const float* gamma_b, const float* stats, void* d_a, float* part, int64_t rows, float drop_p,
                                uint64_t drop_seed, t2s_stream_t stream) {
  T2S_CHECK_ARG(g_out && a && bbox && w_box && b_box && gamma_a && gamma_b && stats && d_a && part, "ocr_tail_bwd: null pointer");
  T2S_CHECK_ARG(rows > 0 && is_dt(a_dtype), "ocr_tail_bwd: bad shape / dtype");
  T2S_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "ocr_tail_bwd: bad dropout");
  const DropCfg drop = make_drop(drop_p, drop_seed);
  dim3 grid(bwd_parts(rows)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (a_dtype == T2S_BF16)
    hipLaunchKernelGGL(ocr_tail_bwd_kernel<bf16_t>, grid, block, 0, st, g_out, (const bf16_t*)a, bbox, w_box, b_box, gamma_a, gamma_b, stats, (bf16_t*)d_a,
                       part, rows, drop);
  else
    hipLaunchKernelGGL(ocr_tail_bwd_kernel<float>, grid, block, 0, st, g_out, (const float*)a, bbox, w_box, b_box, gamma_a, gamma_b, stats, (float*)d_a,
                       part, rows, drop);
  T2S_CHECK_LAUNCH("ocr_tail_bwd");
  return 0;
}

extern "C" int t2s_wide_add_layernorm_fwd(float* h, int64_t h_row_stride, const void* branch, int branch_dtype, const float* bias,
                                          const float* gamma, const float* beta, void* y, int y_dtype, int64_t rows, int width, float eps,
                                          t2s_stream_t stream) {
  T2S_CHECK_ARG(h && gamma && beta && y, "wide_add_layernorm_fwd: null pointer");
  T2S_CHECK_ARG(rows > 0 && is_dt(y_dtype) && (!branch || is_dt(branch_dtype)), "wide_add_layernorm_fwd: bad rows / dtype");
  T2S_CHECK_ARG(width >= 4 && width <= 1280 && width % 4 == 0, "wide_add_layernorm_fwd: width %d is not a multiple of 4 in [4, 1280]", width);
  T2S_CHECK_ARG(h_row_stride >= width && h_row_stride % 4 == 0, "wide_add_layernorm_fwd: bad row stride %lld", (long long)h_row_stride);
  T2S_CHECK_ARG(branch || !bias, "wide_add_layernorm_fwd: a bias without a branch");
  dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int combo = ((branch && branch_dtype == T2S_BF16) ? 2 : 0) + (y_dtype == T2S_BF16 ? 1 : 0);
#define WIDE_LN2(NV_, TB_, TY_)                                                                                                         \
  hipLaunchKernelGGL((wide_add_layernorm_fwd_kernel<NV_, TB_, TY_>), grid, block, 0, st, h, (const TB_*)branch, bias, gamma, beta,      \
                     (TY_*)y, rows, h_row_stride, width, eps)
#define WIDE_LN(NV_)                                                \
  switch (combo) {                                                  \
    case 0: WIDE_LN2(NV_, float, float); break;                     \
    case 1: WIDE_LN2(NV_, float, bf16_t); break;                    \
    case 2: WIDE_LN2(NV_, bf16_t, float); break;                    \
    default: WIDE_LN2(NV_, bf16_t, bf16_t); break;                  \
  }
  const int nv = (width + 255) / 256;
  if (nv == 1) { WIDE_LN(1) } else if (nv == 2) { WIDE_LN(2) } else if (nv == 3) { WIDE_LN(3) } else if (nv == 4) { WIDE_LN(4) } else { WIDE_LN(5) }
#undef WIDE_LN
#undef WIDE_LN2
  T2S_CHECK_LAUNCH("wide_add_layernorm_fwd");
  return 0;
}
