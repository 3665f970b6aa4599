// Which bf16 MFMA shape does the chip run faster on RANDOM operands (the power-limited case)?  Every wave runs N groups of four
// independent accumulations of v_mfma_f32_32x32x16_bf16 (SHAPE 0) or eight of v_mfma_f32_16x16x32_bf16 (SHAPE 1: the same FLOPs
// per group), operands in registers; reports TFLOP/s, shader cycles per group and the clock (s_memtime per s_memrealtime tick).
//   hipcc --offload-arch=gfx950 -O3 tools/ablate/mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ inline unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int SHAPE, bool RANDOM, int FILL, int LDSR = 0>
__global__ __launch_bounds__(256, 2) void probe(float* out, unsigned long long* t, int n) {
  __shared__ __attribute__((aligned(16))) unsigned lds[LDSR ? 16384 : 4];      // 64 KB of random words
  if (LDSR) {
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = mix(i * 2654435761u + blockIdx.x);
    __syncthreads();
  }
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 lacc = {0, 0, 0, 0};
  bf16x8 a[2], b[2];
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 8; ++j) {
      const unsigned h = mix((blockIdx.x * 256 + threadIdx.x) * 16 + k * 8 + j);
      a[k][j] = RANDOM ? (__bf16)(((int)(h & 0xFFFF) - 32768) * (1.0f / 16384.f)) : (__bf16)1.0f;
      b[k][j] = RANDOM ? (__bf16)(((int)(h >> 16) - 32768) * (1.0f / 16384.f)) : (__bf16)0.5f;
    }
  f32x16 c[4] = {};
  f32x4 d[8] = {};
  float e[8];
  for (int j = 0; j < 8; ++j) e[j] = (threadIdx.x & 63) * 1e-3f + j * 0.1f;
  unsigned long long m0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    if (SHAPE == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 1], b[(u >> 1) & 1], c[u], 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u & 1], b[(u >> 1) & 1], d[u], 0, 0, 0);
    }
    // LDSR: ds_read_b128 per group (conflict-free: consecutive lanes, consecutive 16-byte chunks), results kept alive
#pragma unroll
    for (int j = 0; j < LDSR; ++j) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(&lds[((threadIdx.x * 4 + j * 1024 + i * 64) & 16383) & ~3]);
      lacc ^= v;
    }
    // FILL: softmax-like VALU work beside the MFMAs (per group: FILL x (v_exp + v_fma + v_cvt-like mul)), values stay bounded
#pragma unroll
    for (int j = 0; j < FILL; ++j) {
      e[j & 7] = __builtin_amdgcn_exp2f(e[j & 7] * -0.5f);
      e[(j + 3) & 7] = __builtin_fmaf(e[j & 7], 0.25f, e[(j + 3) & 7] * 0.5f);
    }
  }
  unsigned long long m1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int u = 0; u < 4; ++u) for (int j = 0; j < 16; ++j) s += c[u][j];
  for (int u = 0; u < 8; ++u) for (int j = 0; j < 4; ++j) s += d[u][j];
  for (int j = 0; j < 8; ++j) s += e[j];
  s += (float)(lacc[0] ^ lacc[1] ^ lacc[2] ^ lacc[3]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = m1 - m0; t[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int SHAPE, bool RANDOM, int FILL = 0, int LDSR = 0>
void run(int blocks, int n) {
  float* out; unsigned long long* t;
  hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&t, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<SHAPE, RANDOM, FILL, LDSR>), dim3(blocks), dim3(256), 0, 0, out, t, n);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL((probe<SHAPE, RANDOM, FILL, LDSR>), dim3(blocks), dim3(256), 0, 0, out, t, n); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks); hipMemcpy(h.data(), t, blocks * 16, hipMemcpyDeviceToHost);
  double mc = 0, rc = 0; for (int i = 0; i < blocks; ++i) { mc += h[2 * i]; rc += h[2 * i + 1]; }
  const double waves = (double)blocks * 4, fl = waves * n * 4.0 * 32 * 32 * 16 * 2;
  printf("lds %2d, fill %2d, %s operands, %s, %d waves/SIMD: %8.3f ms  %7.0f TFLOP/s  %6.1f cycles per group of 131 kFLOP/wave  clock %.2f x the realtime tick\n",
         LDSR, FILL, RANDOM ? "random " : "constant", SHAPE ? "16x16x32" : "32x32x16", blocks / 256, ms, fl / (ms * 1e-3) / 1e12, mc / blocks / n, mc / rc);
  hipFree(out); hipFree(t);
}
int main() {
  const int n = 40000;
  for (int rep = 0; rep < 2; ++rep) {
    run<0, false>(512, n); run<1, false>(512, n);
    run<0, true>(512, n);  run<1, true>(512, n);
    run<0, true>(256, n);  run<1, true>(256, n);
    // with VALU work beside the MFMAs (an attention-like duty cycle: 4 x 32 MFMA cycles per group against 16 / 24 x ~16 VALU cycles)
    run<0, true, 16>(256, n / 2); run<1, true, 16>(256, n / 2);
    run<0, true, 24>(256, n / 2); run<1, true, 24>(256, n / 2);
    run<0, true, 16>(512, n / 2); run<1, true, 16>(512, n / 2);
    // ... and with LDS reads beside them (per group of 4 MFMAs: 8 / 16 ds_read_b128 = 8 / 16 KB per wave)
    run<0, true, 16, 8>(256, n / 2); run<0, true, 16, 16>(256, n / 2); run<0, true, 0, 8>(256, n / 2); run<0, true, 0, 16>(256, n / 2);
  }
  return 0;
}
