// Shared helpers for the gfx950 kernels of libt2s_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/t2s_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define T2S_HIDDEN 768
#define T2S_WAVE 64

// thread-local error slot (t2s_last_error)
void t2s_set_error(const char* fmt, ...);

#define T2S_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      t2s_set_error(__VA_ARGS__);         \
      return 1;                           \
    }                                     \
  } while (0)

#define T2S_CHECK_LAUNCH(name)                                               \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      t2s_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
      return 2;                                                              \
    }                                                                        \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }

// 4 consecutive elements <-> float4, for both storage types
template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <>
struct Vec4<bf16_t> {
  static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r = {(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    return r;
  }
  static __device__ __forceinline__ void store(bf16_t* p, f32x4 v) {
    bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = t;
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
