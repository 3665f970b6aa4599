// PHOC descriptor of OCR tokens on the GPU: normalised token bytes -> the 604 0/1 features the model consumes as
// context_feature_1 (reference: PhocProcessor, pythia/datasets/processors.py:904-928 -> build_phoc.py:9-14 -> the C
// extension pythia/utils/phoc/src/cphoc.c:12-117).  With this kernel the host ships <= `width` bytes per OCR token instead
// of a 2416-byte fp32 row (1.55 GB -> 20 MB per B=64 batch at 100 x 100 OCR tokens).
//
// One wave per token, a character per lane.  A symbol occupying [i/n, (i+1)/n) of the word sets, for every pyramid level
// L in 2..5, the bit of each region [r/L, (r+1)/L) that covers at least half of it; the 50 listed bigrams do the same
// at level 2.  The ">= 0.5" tests are evaluated in IEEE binary32 with correctly rounded divisions, operation for
// operation as cphoc.c:37-38,57-62,97-104 - the rounding of i/n decides border cases, so this is bit-exact or wrong.
// Bits are collected in a per-wave 19-word LDS mask and expanded to a coalesced fp32 row (HBM-write bound:
// 2416 B per token out, <= width B in).
#include "common.h"

namespace {

constexpr int PHOC_DIM = 604;
constexpr int N_UNI = 36;
constexpr int N_BI = 50;

// cphoc.c:32 (data: the order defines the output columns); two characters per 16-bit entry
__constant__ unsigned short BIGRAMS[N_BI] = {
#define BG(a, b) (unsigned short)((a) | ((b) << 8))
    BG('t', 'h'), BG('h', 'e'), BG('i', 'n'), BG('e', 'r'), BG('a', 'n'), BG('r', 'e'), BG('e', 's'), BG('o', 'n'), BG('s', 't'), BG('n', 't'),
    BG('e', 'n'), BG('a', 't'), BG('e', 'd'), BG('n', 'd'), BG('t', 'o'), BG('o', 'r'), BG('e', 'a'), BG('t', 'i'), BG('a', 'r'), BG('t', 'e'),
    BG('n', 'g'), BG('a', 'l'), BG('i', 't'), BG('a', 's'), BG('i', 's'), BG('h', 'a'), BG('e', 't'), BG('s', 'e'), BG('o', 'u'), BG('o', 'f'),
    BG('l', 'e'), BG('s', 'a'), BG('v', 'e'), BG('r', 'o'), BG('r', 'a'), BG('r', 'i'), BG('h', 'i'), BG('n', 'e'), BG('m', 'e'), BG('d', 'e'),
    BG('c', 'o'), BG('t', 'a'), BG('e', 'c'), BG('s', 'i'), BG('l', 'l'), BG('s', 'o'), BG('n', 'a'), BG('l', 'i'), BG('l', 'a'), BG('e', 'l')
#undef BG
};

__device__ __forceinline__ bool covers_half(float occ0, float occ1, int region, int level) {
  const float r0 = __fdiv_rn((float)region, (float)level), r1 = __fdiv_rn((float)(region + 1), (float)level);
  const float o0 = occ0 > r0 ? occ0 : r0, o1 = occ1 < r1 ? occ1 : r1;
  return __fdiv_rn(__fsub_rn(o1, o0), __fsub_rn(occ1, occ0)) >= 0.5f;
}

template <bool VEC4>
__global__ __launch_bounds__(256) void phoc_kernel(const uint8_t* __restrict__ tokens, int64_t n_tokens, int width,
                                                   float* __restrict__ out, int64_t out_rs) {
  __shared__ uint32_t mask_s[4][20];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + wave;
  if (tok >= n_tokens) return;                       // wave-uniform
  uint32_t* mask = mask_s[wave];
  if (lane < 20) mask[lane] = 0u;
  const uint8_t* w = tokens + tok * width;
  // length = number of bytes before the first NUL (slots are NUL padded)
  int n = 0;
  for (int c0 = 0; c0 < width; c0 += 64) {
    const int i = c0 + lane;
    const bool nz = i < width && w[i] != 0;
    const unsigned long long b = __ballot(nz);
    const int lead = b == ~0ull ? 64 : __ffsll((long long)~b) - 1;      // leading non-zero lanes of this chunk
    n += lead;
    if (lead < 64) break;
  }
  bool bad = false;
  const float fn = (float)n;
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int i = c0 + lane;
    if (i < n) {
      const uint8_t ch = w[i];
      int ci = -1;
      if (ch >= 'a' && ch <= 'z') ci = ch - 'a';
      else if (ch >= '0' && ch <= '9') ci = 26 + (ch - '0');
      if (ci < 0) {
        bad = true;
      } else {
        const float occ0 = __fdiv_rn((float)i, fn), occ1 = __fdiv_rn((float)(i + 1), fn);
        int base = 0;
#pragma unroll
        for (int level = 2; level <= 5; ++level) {
#pragma unroll
          for (int region = 0; region < level; ++region)
            if (covers_half(occ0, occ1, region, level)) {
              const int bit = (base + region) * N_UNI + ci;
              atomicOr(&mask[bit >> 5], 1u << (bit & 31));
            }
          base += level;
        }
        if (i + 1 < n) {
          const unsigned short pair = (unsigned short)(ch | (w[i + 1] << 8));
          int bi = -1;
          for (int k = 0; k < N_BI; ++k)
            if (BIGRAMS[k] == pair) { bi = k; break; }
          if (bi >= 0) {
            const float b0 = __fdiv_rn((float)i, fn), b1 = __fdiv_rn((float)(i + 2), fn);
#pragma unroll
            for (int region = 0; region < 2; ++region)
              if (covers_half(b0, b1, region, 2)) {
                const int bit = 14 * N_UNI + region * N_BI + bi;
                atomicOr(&mask[bit >> 5], 1u << (bit & 31));
              }
          }
        }
      }
    }
  }
  const bool any_bad = __any(bad);
  // (LDS atomics of this wave are complete before its own later LDS reads: same wave, in-order DS queue)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  float* o = out + tok * out_rs;
  const float poison = __builtin_nanf("");            // a symbol outside [a-z0-9] (the reference raises): poison the row
  if (VEC4) {                                         // 604 = 151 x 4: 16-byte stores, 1 KiB per wave instruction
    for (int j4 = lane; j4 < PHOC_DIM / 4; j4 += 64) {
      const int j = 4 * j4;
      const uint32_t bits = (uint32_t)((((uint64_t)mask[(j >> 5) + 1] << 32) | mask[j >> 5]) >> (j & 31));
      f32x4 v = {bits & 1u ? 1.f : 0.f, bits & 2u ? 1.f : 0.f, bits & 4u ? 1.f : 0.f, bits & 8u ? 1.f : 0.f};
      if (any_bad) v = f32x4{poison, poison, poison, poison};
      *reinterpret_cast<f32x4*>(o + j) = v;
    }
  } else {
    for (int j = lane; j < PHOC_DIM; j += 64) {
      const float v = (mask[j >> 5] >> (j & 31)) & 1u ? 1.f : 0.f;
      o[j] = any_bad ? poison : v;
    }
  }
}

}  // namespace

extern "C" int t2s_phoc(const uint8_t* tokens, int64_t n_tokens, int width, float* out, int64_t out_row_stride, t2s_stream_t stream) {
  T2S_CHECK_ARG(n_tokens >= 0 && width > 0 && width <= 4096, "t2s_phoc: n_tokens=%lld width=%d", (long long)n_tokens, width);
  T2S_CHECK_ARG(out_row_stride >= PHOC_DIM, "t2s_phoc: out_row_stride %lld < 604", (long long)out_row_stride);
  if (n_tokens == 0) return 0;
  T2S_CHECK_ARG(tokens && out, "t2s_phoc: null pointer");
  const dim3 grid((unsigned)((n_tokens + 3) / 4)), block(256);
  if (out_row_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
    hipLaunchKernelGGL(phoc_kernel<true>, grid, block, 0, (hipStream_t)stream, tokens, n_tokens, width, out, out_row_stride);
  else
    hipLaunchKernelGGL(phoc_kernel<false>, grid, block, 0, (hipStream_t)stream, tokens, n_tokens, width, out, out_row_stride);
  T2S_CHECK_LAUNCH("phoc");
  return 0;
}
