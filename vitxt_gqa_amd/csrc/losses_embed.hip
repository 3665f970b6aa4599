// HBM-bound row kernels at the two ends of the T2S path for gfx950:
//
//  * t2s_embed_rows      T2S._forward_obj_encoding / _forward_ocr_encoding (pythia/models/t2s.py:192-258) up to
//                        the Linear: out[row] = [ L2norm(f0[row]) | L2norm(f1[row]) | emb0[id0[row]] | emb1[id1[row]] ]
//                        (F.normalize: x / max(||x||, 1e-12)).  One wavefront per row, coalesced 16-B feature loads,
//                        the concatenated GEMM input row is written once in the GEMM operand dtype.
//                        Algorithmic bytes per row: (d0 + d1) * 4 read + (d0 + d1 + 50 k) * sizeof(out) written.
//  * t2s_bce_masked      POSBCEWithMaskLoss.forward (pythia/modules/losses.py:329-343): per-row sums of
//                        BCEWithLogits(x, t) * mask[row] and, in the same pass, the unscaled gradient
//                        (sigmoid(x) - t) * mask[row].
//  * t2s_infonce_stats / t2s_infonce_bwd   InfoNCE.forward (losses.py:361-385): the five per-row bilinear
//                        statistics (q.q, p.p, n.n, q.p, q.n) of the ref / pos / neg logits in ONE pass over the
//                        three [B*12, V+N] tensors; the scalar loss is assembled from them on [B, 12] tensors and the
//                        backward is a second single pass dq = 2a q + d p + e n, dp = 2b p + d q, dn = 2c n + e q.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

template <typename TO>
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ f0, int d0, const float* __restrict__ f1, int d1,
                                                         const int64_t* __restrict__ id0, const float* __restrict__ emb0,
                                                         const int64_t* __restrict__ id1, const float* __restrict__ emb1,
                                                         int emb_dim, int emb_rows, TO* __restrict__ out, int ld_out, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  TO* o = out + row * ld_out;
  int col = 0;
  for (int part = 0; part < 2; ++part) {
    const float* f = part == 0 ? f0 : f1;
    const int d = part == 0 ? d0 : d1;
    if (!f || d == 0) continue;
    const float* fr = f + row * d;
    const int n4 = d >> 2;                       // d is a multiple of 4 (checked on the host)
    float ss = 0.f;
    for (int i = lane; i < n4; i += 64) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(fr + i * 4);
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ss = wave_sum(ss);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    for (int i = lane; i < n4; i += 64) {          // second read hits L1/L2 (row <= 4 KiB)
      const f32x4 v = *reinterpret_cast<const f32x4*>(fr + i * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[col + i * 4 + j] = (TO)(v[j] * inv);
    }
    col += d;
  }
  for (int part = 0; part < 2; ++part) {
    const int64_t* ids = part == 0 ? id0 : id1;
    const float* emb = part == 0 ? emb0 : emb1;
    if (!ids) continue;
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= emb_rows ? emb_rows - 1 : id);       // nn.Embedding would raise; clamp instead of faulting
    if (lane < emb_dim) o[col + lane] = (TO)emb[id * emb_dim + lane];
    col += emb_dim;
  }
}

// one workgroup per row of C logits
__global__ __launch_bounds__(256) void bce_masked_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                         const float* __restrict__ row_mask, float* __restrict__ row_loss,
                                                         float* __restrict__ grad, int C) {
  __shared__ float sh[4];
  const int64_t r = blockIdx.x;
  const float m = row_mask[r];
  const float* xr = x + r * C;
  const float* tr = t + r * C;
  float* gr = grad + r * C;
  float s = 0.f;
  for (int i = threadIdx.x; i < C; i += 256) {
    const float xv = xr[i], tv = tr[i];
    // max(x,0) - x*t + log1p(exp(-|x|))
    s += fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
    gr[i] = (1.f / (1.f + expf(-xv)) - tv) * m;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) row_loss[r] = s * m;
}

__global__ __launch_bounds__(256) void infonce_stats_kernel(const float* __restrict__ q, const float* __restrict__ p,
                                                            const float* __restrict__ n, float* __restrict__ stats, int C) {
  __shared__ float sh[4];
  const int64_t r = blockIdx.x;
  const float* qr = q + r * C;
  const float* pr = p + r * C;
  const float* nr = n + r * C;
  float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < C; i += 256) {
    const float qv = qr[i], pv = pr[i], nv = nr[i];
    a[0] += qv * qv; a[1] += pv * pv; a[2] += nv * nv; a[3] += qv * pv; a[4] += qv * nv;
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float s = block_sum(a[j], sh);
    if (threadIdx.x == 0) stats[r * 5 + j] = s;
  }
}

__global__ __launch_bounds__(256) void infonce_bwd_kernel(const float* __restrict__ q, const float* __restrict__ p,
                                                          const float* __restrict__ n, const float* __restrict__ g,
                                                          float* __restrict__ dq, float* __restrict__ dp, float* __restrict__ dn, int C) {
  const int64_t r = blockIdx.x;
  const float ga = 2.f * g[r * 5 + 0], gb = 2.f * g[r * 5 + 1], gc = 2.f * g[r * 5 + 2], gd = g[r * 5 + 3], ge = g[r * 5 + 4];
  for (int i = threadIdx.x; i < C; i += 256) {
    const float qv = q[r * C + i], pv = p[r * C + i], nv = n[r * C + i];
    dq[r * C + i] = ga * qv + gd * pv + ge * nv;
    dp[r * C + i] = gb * pv + gd * qv;
    dn[r * C + i] = gc * nv + ge * qv;
  }
}

}  // namespace

extern "C" int t2s_embed_rows(const float* f0, int d0, const float* f1, int d1, const int64_t* id0, const float* emb0,
                              const int64_t* id1, const float* emb1, int emb_dim, int emb_rows, void* out, int ld_out,
                              int64_t rows, int out_dtype, t2s_stream_t stream) {
  T2S_CHECK_ARG(f0 && out && rows > 0, "embed_rows: null pointer / empty");
  T2S_CHECK_ARG(d0 > 0 && d0 % 4 == 0 && d1 >= 0 && d1 % 4 == 0, "embed_rows: feature widths must be multiples of 4 (got %d, %d)", d0, d1);
  T2S_CHECK_ARG((f1 != nullptr) == (d1 > 0), "embed_rows: f1 / d1 mismatch");
  T2S_CHECK_ARG((id0 == nullptr) == (emb0 == nullptr) && (id1 == nullptr) == (emb1 == nullptr), "embed_rows: ids / tables mismatch");
  T2S_CHECK_ARG(emb_dim >= 0 && emb_dim <= 64 && emb_rows > 0, "embed_rows: embedding width must be <= 64");
  const int need = d0 + d1 + (id0 ? emb_dim : 0) + (id1 ? emb_dim : 0);
  T2S_CHECK_ARG(ld_out >= need, "embed_rows: output row stride %d < %d", ld_out, need);
  T2S_CHECK_ARG(out_dtype == T2S_F32 || out_dtype == T2S_BF16, "embed_rows: bad dtype %d", out_dtype);
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == T2S_BF16)
    hipLaunchKernelGGL(embed_rows_kernel<bf16_t>, grid, block, 0, st, f0, d0, f1, d1, id0, emb0, id1, emb1, emb_dim, emb_rows, (bf16_t*)out, ld_out, rows);
  else
    hipLaunchKernelGGL(embed_rows_kernel<float>, grid, block, 0, st, f0, d0, f1, d1, id0, emb0, id1, emb1, emb_dim, emb_rows, (float*)out, ld_out, rows);
  T2S_CHECK_LAUNCH("embed_rows");
  return 0;
}

extern "C" int t2s_bce_masked(const float* scores, const float* targets, const float* row_mask, float* row_loss, float* grad,
                              int64_t rows, int cols, t2s_stream_t stream) {
  T2S_CHECK_ARG(scores && targets && row_mask && row_loss && grad, "bce_masked: null pointer");
  T2S_CHECK_ARG(rows > 0 && rows < ((int64_t)1 << 31) && cols > 0, "bce_masked: bad shape");
  hipLaunchKernelGGL(bce_masked_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, scores, targets, row_mask, row_loss, grad, cols);
  T2S_CHECK_LAUNCH("bce_masked");
  return 0;
}

extern "C" int t2s_infonce_stats(const float* q, const float* p, const float* n, float* stats, int64_t rows, int cols, t2s_stream_t stream) {
  T2S_CHECK_ARG(q && p && n && stats, "infonce_stats: null pointer");
  T2S_CHECK_ARG(rows > 0 && rows < ((int64_t)1 << 31) && cols > 0, "infonce_stats: bad shape");
  hipLaunchKernelGGL(infonce_stats_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, q, p, n, stats, cols);
  T2S_CHECK_LAUNCH("infonce_stats");
  return 0;
}

extern "C" int t2s_infonce_bwd(const float* q, const float* p, const float* n, const float* gstats, float* dq, float* dp, float* dn,
                               int64_t rows, int cols, t2s_stream_t stream) {
  T2S_CHECK_ARG(q && p && n && gstats && dq && dp && dn, "infonce_bwd: null pointer");
  T2S_CHECK_ARG(rows > 0 && rows < ((int64_t)1 << 31) && cols > 0, "infonce_bwd: bad shape");
  hipLaunchKernelGGL(infonce_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, q, p, n, gstats, dq, dp, dn, cols);
  T2S_CHECK_LAUNCH("infonce_bwd");
  return 0;
}
