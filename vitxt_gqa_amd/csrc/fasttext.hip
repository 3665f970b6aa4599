// FastText token vectors from a table resident in HBM (gfx950; HBM-bound gather).
//
// Replaces the host-side lookup of FastTextProcessor (pythia/datasets/processors.py:478-491) -> WordToVectorDict
// (pythia/utils/vocab.py:375-381) -> third-party fasttext FastText::getWordVector: for every OCR token slot
//     token = mean over its space-separated words of  word = (sum of the word's subword rows) * float(1 / #rows)
// in exactly that order of fp32 operations (rows added in id order, one multiply per word, the words summed in order and
// divided by their count), so the result is bit-equal to the library's.  One wavefront per token slot; lane l owns the float4
// chunks l, l + 64, ... of the row (dim 300: 75 chunks); the subword ids of a slot are read once through the wave (scalar
// broadcast) and each table row is fetched as coalesced 16-byte loads.  Algorithmic bytes per slot: #rows * dim * 4 read (random
// 1.2-KB rows: L2 / Infinity-Cache / HBM by table size) + dim * 4 written.
#include "common.h"

namespace {

constexpr int FT_MAX_CHUNKS = 4;       // float4 chunks per lane: dims up to 1024

__global__ __launch_bounds__(256) void fasttext_rows_kernel(const float* __restrict__ table, int64_t table_rows, int dim, const int32_t* __restrict__ ids,
                                                            const uint8_t* __restrict__ word_end, const int32_t* __restrict__ off, int64_t slots,
                                                            float* __restrict__ out) {
#pragma clang fp contract(off)          // multiply and add round separately, as the library's Vector::mul / numpy's sum do (no FMA)
  const int lane = threadIdx.x & 63;
  const int64_t slot = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= slots) return;
  const int nch = dim >> 2;
  f32x4 word[FT_MAX_CHUNKS], tok[FT_MAX_CHUNKS];
#pragma unroll
  for (int c = 0; c < FT_MAX_CHUNKS; ++c) { word[c] = f32x4{0.f, 0.f, 0.f, 0.f}; tok[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int e0 = off[slot], e1 = off[slot + 1];
  int rows_in_word = 0, words = 0;
  for (int e = e0; e < e1; ++e) {
    const int id = ids[e];                           // same address in every lane: one broadcast load
    if (id >= 0 && id < table_rows) {
      const float* row = table + (int64_t)id * dim;
#pragma unroll
      for (int c = 0; c < FT_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * ch);
          word[c] = word[c] + v;
        }
      }
      ++rows_in_word;
    }
    if (word_end[e]) {
      const float inv = rows_in_word ? (float)(1.0 / (double)rows_in_word) : 0.f;     // real(1.0 / ngrams.size())
#pragma unroll
      for (int c = 0; c < FT_MAX_CHUNKS; ++c) {
        tok[c] = tok[c] + word[c] * inv;
        word[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      rows_in_word = 0;
      ++words;
    }
  }
  float* o = out + slot * dim;
#pragma unroll
  for (int c = 0; c < FT_MAX_CHUNKS; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      f32x4 r = tok[c];
      if (words > 1) r = r / (float)words;           // np.mean over the words of a token (a single word: sum / 1 is exact)
      *reinterpret_cast<f32x4*>(o + 4 * ch) = r;
    }
  }
}

}  // namespace

extern "C" int t2s_fasttext_rows(const float* table, int64_t table_rows, int dim, const int32_t* ids, const uint8_t* word_end,
                                 const int32_t* offsets, int64_t slots, float* out, t2s_stream_t stream) {
  T2S_CHECK_ARG(table && ids && word_end && offsets && out, "fasttext_rows: null pointer");
  T2S_CHECK_ARG(table_rows > 0 && slots > 0 && dim > 0 && dim % 4 == 0 && dim <= 256 * FT_MAX_CHUNKS,
                "fasttext_rows: dim %d must be a multiple of 4 and at most %d", dim, 256 * FT_MAX_CHUNKS);
  hipLaunchKernelGGL(fasttext_rows_kernel, dim3((unsigned)((slots + 3) / 4)), dim3(256), 0, (hipStream_t)stream, table, table_rows, dim, ids, word_end,
                     offsets, slots, out);
  T2S_CHECK_LAUNCH("fasttext_rows");
  return 0;
}
