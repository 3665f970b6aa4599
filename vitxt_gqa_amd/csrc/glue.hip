// Elementwise passes around the encoders that the reference spells as chains of framework ops, one HBM pass each here.
//   * QTV's modality residual  x = x + tanh(enc_out)  (pythia/models/t2s.py:428-432), forward and backward;
//   * "fp32 tensor + operand-dtype tensor -> fp32" (the residual path of a gradient meeting the bf16 input gradient of an encoder).
// Tensors are [B, rows, 768] row-major; where a pointer is paired with a batch stride (in elements) the B blocks of rows may
// sit inside a larger buffer (a gradient that arrives as a slice of the next stage's input gradient).  HBM-bound: 4 elements per
// lane per access (16-B fp32 / 8-B bf16 vectors); algorithmic bytes per element: fwd 4+4 read, 4 written; bwd 4+4 read, 2 (or 4)
// written; add_cast 4+2 read, 4 written.
#include "common.h"

namespace {

// one workgroup-stride loop over the B * rows * 192 float4 groups; b / r recovered per group (rows * 192 groups per batch)
template <typename F>
__device__ __forceinline__ void for_each_group(int64_t B, int64_t per_b4, F f) {
  const int64_t n4 = B * per_b4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / per_b4;
    f(b, (i - b * per_b4) * 4, i * 4);          // (sample, element offset inside the sample, element offset in a contiguous tensor)
  }
}

__global__ __launch_bounds__(256) void tanh_residual_fwd_kernel(const float* __restrict__ x, const float* __restrict__ o, float* __restrict__ y,
                                                                int64_t B, int64_t per_b4, int64_t y_bs) {
  for_each_group(B, per_b4, [&](int64_t b, int64_t e, int64_t c) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + c), ov = *reinterpret_cast<const f32x4*>(o + c);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = xv[j] + tanhf(ov[j]);
    *reinterpret_cast<f32x4*>(y + b * y_bs + e) = r;
  });
}

template <typename TO>
__global__ __launch_bounds__(256) void tanh_residual_bwd_kernel(const float* __restrict__ gy, int64_t gy_bs, const float* __restrict__ o,
                                                                TO* __restrict__ go, int64_t B, int64_t per_b4) {
  for_each_group(B, per_b4, [&](int64_t b, int64_t e, int64_t c) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gy + b * gy_bs + e), ov = *reinterpret_cast<const f32x4*>(o + c);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = tanhf(ov[j]);
      r[j] = g[j] * (1.f - t * t);
    }
    Vec4<TO>::store(go + c, r);
  });
}

template <typename TB>
__global__ __launch_bounds__(256) void add_cast_kernel(const float* __restrict__ a, int64_t a_bs, const TB* __restrict__ bb, float* __restrict__ out,
                                                       int64_t B, int64_t per_b4) {
  for_each_group(B, per_b4, [&](int64_t b, int64_t e, int64_t c) {
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + b * a_bs + e), bv = Vec4<TB>::load(bb + c);
    f32x4 r = {av[0] + bv[0], av[1] + bv[1], av[2] + bv[2], av[3] + bv[3]};
    *reinterpret_cast<f32x4*>(out + c) = r;
  });
}

inline unsigned glue_grid(int64_t n4) {
  int64_t blocks = (n4 + 255) / 256;
  return (unsigned)(blocks > 16384 ? 16384 : blocks);
}

}  // namespace

extern "C" int t2s_tanh_residual_fwd(const float* x, const float* enc_out, float* y, int64_t B, int64_t rows, int64_t y_batch_stride,
                                     t2s_stream_t stream) {
  T2S_CHECK_ARG(x && enc_out && y, "tanh_residual_fwd: null pointer");
  T2S_CHECK_ARG(B > 0 && rows > 0 && y_batch_stride >= rows * T2S_HIDDEN && y_batch_stride % 4 == 0, "tanh_residual_fwd: bad shape");
  const int64_t per_b4 = rows * (T2S_HIDDEN / 4);
  hipLaunchKernelGGL(tanh_residual_fwd_kernel, dim3(glue_grid(B * per_b4)), dim3(256), 0, (hipStream_t)stream, x, enc_out, y, B, per_b4, y_batch_stride);
  T2S_CHECK_LAUNCH("tanh_residual_fwd");
  return 0;
}

extern "C" int t2s_tanh_residual_bwd(const float* gy, int64_t gy_batch_stride, const float* enc_out, void* g_enc, int g_dtype, int64_t B,
                                     int64_t rows, t2s_stream_t stream) {
  T2S_CHECK_ARG(gy && enc_out && g_enc, "tanh_residual_bwd: null pointer");
  T2S_CHECK_ARG(B > 0 && rows > 0 && gy_batch_stride >= rows * T2S_HIDDEN && gy_batch_stride % 4 == 0, "tanh_residual_bwd: bad shape");
  T2S_CHECK_ARG(g_dtype == T2S_F32 || g_dtype == T2S_BF16, "tanh_residual_bwd: bad dtype %d", g_dtype);
  const int64_t per_b4 = rows * (T2S_HIDDEN / 4);
  dim3 grid(glue_grid(B * per_b4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (g_dtype == T2S_BF16) hipLaunchKernelGGL(tanh_residual_bwd_kernel<bf16_t>, grid, block, 0, st, gy, gy_batch_stride, enc_out, (bf16_t*)g_enc, B, per_b4);
  else hipLaunchKernelGGL(tanh_residual_bwd_kernel<float>, grid, block, 0, st, gy, gy_batch_stride, enc_out, (float*)g_enc, B, per_b4);
  T2S_CHECK_LAUNCH("tanh_residual_bwd");
  return 0;
}

extern "C" int t2s_add_cast(const float* a, int64_t a_batch_stride, const void* b, int b_dtype, float* out, int64_t B, int64_t rows,
                            t2s_stream_t stream) {
  T2S_CHECK_ARG(a && b && out, "add_cast: null pointer");
  T2S_CHECK_ARG(B > 0 && rows > 0 && a_batch_stride >= rows * T2S_HIDDEN && a_batch_stride % 4 == 0, "add_cast: bad shape");
  T2S_CHECK_ARG(b_dtype == T2S_F32 || b_dtype == T2S_BF16, "add_cast: bad dtype %d", b_dtype);
  const int64_t per_b4 = rows * (T2S_HIDDEN / 4);
  dim3 grid(glue_grid(B * per_b4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (b_dtype == T2S_BF16) hipLaunchKernelGGL(add_cast_kernel<bf16_t>, grid, block, 0, st, a, a_batch_stride, (const bf16_t*)b, out, B, per_b4);
  else hipLaunchKernelGGL(add_cast_kernel<float>, grid, block, 0, st, a, a_batch_stride, (const float*)b, out, B, per_b4);
  T2S_CHECK_LAUNCH("add_cast");
  return 0;
}
