// GELU (erf form) of the third-party BertIntermediate (hidden_act="gelu"), forward and backward.
// HBM-bound elementwise passes: 4 elements per lane per access (8-B bf16 / 16-B fp32 vectors).
// Backward also produces per-column partial sums of du (= gradient of the 3072-wide FFN bias) so
// the [rows, 3072] tensor is read once: each workgroup owns a 1024-column stripe, walks rows with a
// grid stride and writes one partial row.
// Algorithmic bytes per element: fwd 2*sizeof(T); bwd 3*sizeof(T).
#include "common.h"

namespace {

constexpr int GELU_MAX_PARTS = 1024;

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ u, T* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = Vec4<T>::load(u + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = gelu_f(v[j]);
    Vec4<T>::store(y + i * 4, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ u, T* __restrict__ du,
                                                       float* __restrict__ dbias_part, int64_t rows, int cols) {
  const int col = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (col >= cols) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const f32x4 g = Vec4<T>::load(dy + r * cols + col);
    const f32x4 x = Vec4<T>::load(u + r * cols + col);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = g[j] * gelu_grad_f(x[j]);
      acc[j] += o[j];
    }
    Vec4<T>::store(du + r * cols + col, o);
  }
  *reinterpret_cast<f32x4*>(dbias_part + (int64_t)blockIdx.x * cols + col) = acc;
}

int gelu_parts(int64_t rows) { return (int)(rows < GELU_MAX_PARTS ? (rows < 1 ? 1 : rows) : GELU_MAX_PARTS); }

}  // namespace

extern "C" int t2s_gelu_fwd(const void* u, void* y, int64_t n, int dtype, t2s_stream_t stream) {
  T2S_CHECK_ARG(u && y, "gelu_fwd: null pointer");
  T2S_CHECK_ARG(n > 0 && n % 4 == 0, "gelu_fwd: element count %lld must be a positive multiple of 4", (long long)n);
  T2S_CHECK_ARG(dtype == T2S_F32 || dtype == T2S_BF16, "gelu_fwd: bad dtype %d", dtype);
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == T2S_BF16)
    hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)u, (bf16_t*)y, n4);
  else
    hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)u, (float*)y, n4);
  T2S_CHECK_LAUNCH("gelu_fwd");
  return 0;
}

extern "C" int t2s_gelu_bwd_parts(int64_t rows) { return gelu_parts(rows); }

extern "C" int t2s_gelu_bwd(const void* dy, const void* u, void* du, float* dbias_part, int64_t rows, int cols, int dtype,
                            t2s_stream_t stream) {
  T2S_CHECK_ARG(dy && u && du && dbias_part, "gelu_bwd: null pointer");
  T2S_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0, "gelu_bwd: bad shape rows=%lld cols=%d", (long long)rows, cols);
  T2S_CHECK_ARG(dtype == T2S_F32 || dtype == T2S_BF16, "gelu_bwd: bad dtype %d", dtype);
  dim3 grid(gelu_parts(rows), (cols / 4 + 255) / 256), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == T2S_BF16)
    hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)u, (bf16_t*)du, dbias_part, rows, cols);
  else
    hipLaunchKernelGGL(gelu_bwd_kernel<float>, grid, block, 0, st, (const float*)dy, (const float*)u, (float*)du, dbias_part, rows, cols);
  T2S_CHECK_LAUNCH("gelu_bwd");
  return 0;
}
