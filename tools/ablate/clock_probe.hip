// clock and MFMA issue-rate probe: every wave runs N back-to-back v_mfma_f32_32x32x16_bf16 (4 independent accumulators),
// optionally with K v_exp_f32 fillers per MFMA; reports core cycles (s_memtime) per MFMA and the core clock (s_memtime / s_memrealtime)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int FILL>
__global__ __launch_bounds__(256, 2) void probe(float* out, unsigned long long* t, int n) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float e[8];
  for (int j = 0; j < 8; ++j) e[j] = threadIdx.x * 1e-3f + j;
  unsigned long long m0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < FILL; ++j) e[j & 7] = __builtin_amdgcn_exp2f(e[j & 7]);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < FILL; ++j) e[(j + 4) & 7] = __builtin_amdgcn_exp2f(e[(j + 4) & 7]);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < FILL; ++j) e[j & 7] = __builtin_amdgcn_exp2f(e[j & 7]);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < FILL; ++j) e[(j + 4) & 7] = __builtin_amdgcn_exp2f(e[(j + 4) & 7]);
  }
  unsigned long long m1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
  for (int j = 0; j < 8; ++j) s += e[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = m1 - m0; t[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int FILL>
void run(int blocks, int threads, int n) {
  float* out; unsigned long long* t;
  hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&t, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<FILL>, dim3(blocks), dim3(threads), 0, 0, out, t, n);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(probe<FILL>, dim3(blocks), dim3(threads), 0, 0, out, t, n); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks); hipMemcpy(h.data(), t, blocks * 16, hipMemcpyDeviceToHost);
  double mc = 0, rc = 0; for (int i = 0; i < blocks; ++i) { mc += h[2 * i]; rc += h[2 * i + 1]; }
  mc /= blocks; rc /= blocks;
  const double waves = (double)blocks * threads / 64, fl = waves * n * 4.0 * 32 * 32 * 16 * 2;
  printf("fill=%d blocks=%d threads=%d: %.3f ms, %.0f TF/s; s_memtime/MFMA/wave %.1f; clock ratio memtime/realtime %.2f (x100MHz?)\n", FILL, blocks, threads, ms,
         fl / (ms * 1e-3) / 1e12, mc / (4.0 * n), mc / rc);
}
int main() {
  run<0>(256 * 2, 256, 20000);   // 2 waves / SIMD
  run<0>(256, 256, 20000);       // 1 wave / SIMD
  run<1>(256 * 2, 256, 20000);
  run<2>(256 * 2, 256, 20000);
  run<3>(256 * 2, 256, 20000);
  run<4>(256 * 2, 256, 20000);
  run<2>(256, 256, 20000);
  run<4>(256, 256, 20000);
  return 0;
}
