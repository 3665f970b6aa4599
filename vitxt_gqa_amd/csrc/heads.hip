// OCR pointer-network scoring head and the grounding scorers / selection for gfx950.
//
//  * t2s_ptr_scores      OcrPtrNet.forward (pythia/models/t2s.py:648-670) + the concat of
//                        T2S._forward_output (:279-286): scores[b, j, n] = q[b,j].k[b,n]/sqrt(768) + mask01[b,n]
//                        (the RAW 0/1 mask is added, Appendix A Q12), written straight into columns
//                        [col0, col0+N) of the [B, 12, V+N] logits buffer.  MFMA 16x16x32 bf16 (12 decoding
//                        steps padded to the 16-row tile), K rows streamed once from HBM: memory-bound,
//                        algorithmic bytes = N*768*sizeof(k) per sample.
//  * t2s_question_pool   Grounding_Module._calculate_self_attn (t2s.py:453-459): softmax over ALL 20 positions,
//                        then mask and renormalise (Q6), then the weighted sum of the projected question.
//  * t2s_attention_score AttentionScore.forward (pythia/modules/spatio_temporal_grounding.py:15-23): unscaled
//                        dot, softmax over all M, mask, renormalise (+1e-12), fill -10000 (Q7).
//  * t2s_ground_select   Temporal_/Spatial_Grounding_Indicator.forward (spatio_temporal_grounding.py:34-68,
//                        79-142) + Grounding_Module.forward t2s.py:486-494: 2-way hard gumbel split with the
//                        exponential draws INJECTED, top-k masks with the build's deterministic tie rule
//                        (lowest index first, Q9), ground_frame, new_ocr_mask, ground boxes.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ------------------------------------------------------------------------------------------------
// pointer scores
constexpr int PTR_QROWS = 16;              // MFMA tile rows (>= number of decoding steps)
constexpr int PTR_LD = T2S_HIDDEN + 8;     // padded bf16 LDS row (16-B pad breaks the 1536-B stride)

template <typename TK>
__device__ __forceinline__ bf16x8 load_k8(const TK* p);
template <>
__device__ __forceinline__ bf16x8 load_k8<bf16_t>(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
template <>
__device__ __forceinline__ bf16x8 load_k8<float>(const float* p) {
  const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 4);
  bf16x8 r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
  return r;
}

// bf16 operands (q rounded to bf16 in LDS), fp32 accumulate.  One wave = 16 keys x 16 query rows.
template <typename TK>
__global__ __launch_bounds__(256) void ptr_scores_mfma_kernel(const float* __restrict__ q, const TK* __restrict__ k,
                                                              const float* __restrict__ mask, float* __restrict__ out,
                                                              int D, int N, int64_t out_row_stride, int col0, float scale) {
  __shared__ __attribute__((aligned(16))) bf16_t qs[PTR_QROWS * PTR_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.y;
  for (int i = tid; i < PTR_QROWS * (T2S_HIDDEN / 4); i += 256) {
    const int r = i / (T2S_HIDDEN / 4), c4 = i % (T2S_HIDDEN / 4);
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    if (r < D) v = *reinterpret_cast<const f32x4_t*>(q + ((int64_t)b * D + r) * T2S_HIDDEN + c4 * 4);
    bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(qs + r * PTR_LD + c4 * 4) = t;
  }
  __syncthreads();
  const int n = blockIdx.x * 64 + wave * 16 + (lane & 15);
  const int kg = lane >> 4;
  const int nc = n < N ? n : N - 1;
  const TK* kp = k + ((int64_t)b * N + nc) * T2S_HIDDEN + kg * 8;
  const bf16_t* qp = qs + (lane & 15) * PTR_LD + kg * 8;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int s = 0; s < T2S_HIDDEN / 32; ++s) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(qp + s * 32);     // A[row = query][k]
    const bf16x8 bb = load_k8<TK>(kp + s * 32);                         // B[k][col = key]
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, acc, 0, 0, 0);
  }
  if (n < N) {
    const float m = mask[(int64_t)b * N + n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = kg * 4 + r;                                          // C/D: col = lane&15, row = (lane>>4)*4 + reg
      if (j < D) out[((int64_t)b * D + j) * out_row_stride + col0 + n] = acc[r] * scale + m;
    }
  }
}

// exact-fp32 variant (parity mode): v_mfma_f32_16x16x4_f32, A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]
__global__ __launch_bounds__(256) void ptr_scores_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ mask, float* __restrict__ out,
                                                             int D, int N, int64_t out_row_stride, int col0, float scale) {
  __shared__ __attribute__((aligned(16))) float qs[PTR_QROWS * (T2S_HIDDEN + 4)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.y;
  for (int i = tid; i < PTR_QROWS * T2S_HIDDEN; i += 256) {
    const int r = i / T2S_HIDDEN, cix = i % T2S_HIDDEN;
    qs[r * (T2S_HIDDEN + 4) + cix] = r < D ? q[((int64_t)b * D + r) * T2S_HIDDEN + cix] : 0.f;
  }
  __syncthreads();
  const int n = blockIdx.x * 64 + wave * 16 + (lane & 15);
  const int kg = lane >> 4;
  const int nc = n < N ? n : N - 1;
  const float* kp = k + ((int64_t)b * N + nc) * T2S_HIDDEN + kg;
  const float* qp = qs + (lane & 15) * (T2S_HIDDEN + 4) + kg;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int s = 0; s < T2S_HIDDEN / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qp[s * 4], kp[s * 4], acc, 0, 0, 0);
  if (n < N) {
    const float m = mask[(int64_t)b * N + n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = kg * 4 + r;
      if (j < D) out[((int64_t)b * D + j) * out_row_stride + col0 + n] = acc[r] * scale + m;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// question pooling: one workgroup per sample; qp [B, T, 768] fp32 (already projected by q_linear)
__global__ __launch_bounds__(256) void question_pool_kernel(const float* __restrict__ qp, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ qmask,
                                                            float* __restrict__ out, int T) {
  __shared__ float att[64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = qp + (int64_t)b * T * T2S_HIDDEN;
  for (int t = wave; t < T; t += 4) {
    float s = 0.f;
    for (int i = lane; i < T2S_HIDDEN; i += 64) s += x[t * T2S_HIDDEN + i] * w[i];
    s = wave_sum(s);
    if (lane == 0) att[t] = s + bias[0];
  }
  __syncthreads();
  if (tid == 0) {
    float mx = -INFINITY;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, att[t]);
    float sum = 0.f;
    for (int t = 0; t < T; ++t) { att[t] = expf(att[t] - mx); sum += att[t]; }
    float msum = 0.f;
    for (int t = 0; t < T; ++t) { att[t] = att[t] / sum * qmask[(int64_t)b * T + t]; msum += att[t]; }
    for (int t = 0; t < T; ++t) att[t] = att[t] / (msum + 1e-12f);
  }
  __syncthreads();
  for (int i = tid; i < T2S_HIDDEN; i += 256) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += att[t] * x[t * T2S_HIDDEN + i];
    out[(int64_t)b * T2S_HIDDEN + i] = s;
  }
}

// dots[b, m] = q[b] . k[b, m]: one wavefront per key row (12 elements per lane, coalesced)
template <typename TK>
__global__ __launch_bounds__(256) void score_dot_kernel(const float* __restrict__ q, const TK* __restrict__ k, int64_t k_bs,
                                                        float* __restrict__ dots, int M, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = (int)(row / M);
  const TK* __restrict__ krow = k + (int64_t)b * k_bs + (row - (int64_t)b * M) * T2S_HIDDEN;       // the M rows of a sample may sit inside a longer sequence
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = (i * 64 + lane) * 4;
    const f32x4 kv = Vec4<TK>::load(krow + e);
    const f32x4 qv = *reinterpret_cast<const f32x4*>(q + (int64_t)b * T2S_HIDDEN + e);
    s += kv[0] * qv[0] + kv[1] * qv[1] + kv[2] * qv[2] + kv[3] * qv[3];
  }
  s = wave_sum(s);
  if (lane == 0) dots[row] = s;
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(v, o, 64);
    v = is_max ? fmaxf(v, t) : v + t;
  }
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
  return r;
}

// in place: dots[b, :] -> masked / renormalised softmax scores (Q7); one workgroup per sample
__global__ __launch_bounds__(256) void score_softmax_kernel(float* __restrict__ sc, const float* __restrict__ mask, int M) {
  __shared__ float sh[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float* s = sc + (int64_t)b * M;
  const float* mk = mask + (int64_t)b * M;
  float mx = -INFINITY;
  for (int i = tid; i < M; i += 256) mx = fmaxf(mx, s[i]);
  mx = block_reduce(mx, true, sh);
  float sum = 0.f;
  for (int i = tid; i < M; i += 256) sum += expf(s[i] - mx);
  sum = block_reduce(sum, false, sh);
  float msum = 0.f;
  for (int i = tid; i < M; i += 256) {
    const float a = expf(s[i] - mx) / sum * mk[i];
    s[i] = a;
    msum += a;
  }
  msum = block_reduce(msum, false, sh);
  for (int i = tid; i < M; i += 256) s[i] = (mk[i] == 0.f) ? -10000.0f : s[i] / (msum + 1e-12f);
}

// ------------------------------------------------------------------------------------------------
// selection.  Shared helper: pick the k best of n values held in LDS (largest or smallest first), ties to the
// lowest index, by k rounds of wave arg-reduction; writes 1.0 into out_mask at the chosen positions.
__device__ __forceinline__ void wave_topk_mask(const float* vals, int n, int k, bool largest, float* out_mask, int lane,
                                                unsigned long long* taken /* per-lane bitset for i = lane + 64*j */) {
  for (int round = 0; round < k && round < n; ++round) {
    float best = largest ? -INFINITY : INFINITY;
    int bi = 0x7fffffff;
    int j = 0;
    for (int i = lane; i < n; i += 64, ++j) {
      if ((*taken >> j) & 1ull) continue;
      const float v = vals[i];
      const bool better = largest ? (v > best) : (v < best);
      if (better || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const bool better = largest ? (ob > best) : (ob < best);
      if (better || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (bi != 0x7fffffff) {
      if ((bi & 63) == lane) {
        *taken |= 1ull << (bi >> 6);
        out_mask[bi] = 1.f;
      }
    }
  }
}

// temporal stage: one wavefront per sample (F <= 4096).  score: [B, F] AttentionScore output; expo: [B, 2, F].
__global__ __launch_bounds__(64) void select_frames_kernel(const float* __restrict__ score, const float* __restrict__ fmask,
                                                           const float* __restrict__ expo, const int64_t* __restrict__ frame_id,
                                                           float* __restrict__ pos_mask, float* __restrict__ neg_mask,
                                                           int64_t* __restrict__ ground_frame, int F, int topk) {
  extern __shared__ float shm[];          // pos_s[F], neg_s[F], pos_top[F], neg_top[F]
  float* pos_s = shm;
  float* neg_s = shm + F;
  float* pos_top = shm + 2 * F;
  float* neg_top = shm + 3 * F;
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < F; i += 64) {
    const float s = score[(int64_t)b * F + i], fm = fmask[(int64_t)b * F + i];
    const float g0 = -logf(expo[((int64_t)b * 2 + 0) * F + i]), g1 = -logf(expo[((int64_t)b * 2 + 1) * F + i]);
    const float pm = ((s + g0) >= (s + g1) ? 1.f : 0.f) * fm;       // argmax over [pos; neg], tie -> pos (Q8)
    const float nm = ((s + g0) >= (s + g1) ? 0.f : 1.f) * fm;
    pos_s[i] = pm == 0.f ? -10000.0f : s * pm;
    neg_s[i] = nm == 0.f ? -10000.0f : s * nm;
    pos_top[i] = 0.f;
    neg_top[i] = 0.f;
  }
  __syncthreads();
  unsigned long long taken = 0;
  wave_topk_mask(pos_s, F, topk, true, pos_top, lane, &taken);
  taken = 0;
  wave_topk_mask(neg_s, F, topk, false, neg_top, lane, &taken);
  __syncthreads();
  for (int i = lane; i < F; i += 64) {
    const float fm = fmask[(int64_t)b * F + i];
    pos_mask[(int64_t)b * F + i] = pos_top[i] * fm;
    neg_mask[(int64_t)b * F + i] = neg_top[i] * fm;
  }
  if (lane == 0) {                          // ground_frame: frame ids of the selected frames, ascending index (Q11)
    int c = 0;
    for (int i = 0; i < F && c < topk; ++i)
      if (pos_top[i] != 0.f) ground_frame[(int64_t)b * topk + c++] = frame_id[(int64_t)b * F + i];
  }
}

// new_ocr_mask[b, n] = any_j (temporal_id[b, n] == max(ground_frame[b, j], 1))      (t2s.py:486-494)
__global__ __launch_bounds__(256) void new_ocr_mask_kernel(const int64_t* __restrict__ temporal_id, const int64_t* __restrict__ ground_frame,
                                                           float* __restrict__ new_mask, int N, int topk) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int64_t t = temporal_id[(int64_t)b * N + n];
  float m = 0.f;
  for (int j = 0; j < topk; ++j) {
    int64_t g = ground_frame[(int64_t)b * topk + j];
    g = g == 0 ? 1 : g;
    if (t == g) m = 1.f;
  }
  new_mask[(int64_t)b * N + n] = m;
}

// spatial stage: one wavefront per (sample, frame) over its P OCR slots (P <= 4096)
__global__ __launch_bounds__(64) void select_ocr_kernel(const float* __restrict__ score, const float* __restrict__ new_mask,
                                                        const float* __restrict__ expo, const float* __restrict__ bbox,
                                                        float* __restrict__ pos_mask, float* __restrict__ neg_mask,
                                                        float* __restrict__ ground_box, int Fn, int P, int topk) {
  extern __shared__ float shm[];          // pos_s[P], neg_s[P], pos_top[P], neg_top[P]
  float* pos_s = shm;
  float* neg_s = shm + P;
  float* pos_top = shm + 2 * P;
  float* neg_top = shm + 3 * P;
  const int f = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int64_t N = (int64_t)Fn * P;
  const int64_t base = (int64_t)b * N + (int64_t)f * P;
  for (int i = lane; i < P; i += 64) {
    const float s = score[base + i], nm_ = new_mask[base + i];
    const float g0 = -logf(expo[((int64_t)b * 2 + 0) * N + (int64_t)f * P + i]);
    const float g1 = -logf(expo[((int64_t)b * 2 + 1) * N + (int64_t)f * P + i]);
    const float pm = ((s + g0) >= (s + g1) ? 1.f : 0.f) * nm_;
    const float nm = ((s + g0) >= (s + g1) ? 0.f : 1.f) * nm_;
    pos_s[i] = pm == 0.f ? -10000.0f : s * pm;
    neg_s[i] = nm == 0.f ? -10000.0f : s * nm;
    pos_top[i] = 0.f;
    neg_top[i] = 0.f;
  }
  __syncthreads();
  unsigned long long taken = 0;
  wave_topk_mask(pos_s, P, topk, true, pos_top, lane, &taken);
  taken = 0;
  wave_topk_mask(neg_s, P, topk, false, neg_top, lane, &taken);
  __syncthreads();
  for (int i = lane; i < P; i += 64) {
    pos_mask[base + i] = pos_top[i];                             // NOT masked by new_mask (the reference's `* attn_mask` is commented out)
    neg_mask[base + i] = neg_top[i] * new_mask[base + i];
  }
  if (lane == 0) {                                               // masked_select order: ascending slot index
    int c = 0;
    for (int i = 0; i < P && c < topk; ++i)
      if (pos_top[i] != 0.f) {
        const float* bx = bbox + (base + i) * 4;
        float* o = ground_box + (((int64_t)b * Fn + f) * topk + c) * 4;
        o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3];
        ++c;
      }
  }
}

}  // namespace

extern "C" int t2s_ptr_scores(const float* q, const void* k, const float* mask, float* out, int B, int D, int N,
                              int64_t out_row_stride, int col0, float scale, int k_dtype, int exact_fp32, t2s_stream_t stream) {
  T2S_CHECK_ARG(q && k && mask && out, "ptr_scores: null pointer");
  T2S_CHECK_ARG(B > 0 && B <= 65535 && D > 0 && D <= PTR_QROWS && N > 0, "ptr_scores: bad shape B=%d D=%d N=%d", B, D, N);
  T2S_CHECK_ARG(col0 >= 0 && out_row_stride >= (int64_t)col0 + N, "ptr_scores: output row too short");
  T2S_CHECK_ARG(k_dtype == T2S_F32 || k_dtype == T2S_BF16, "ptr_scores: bad dtype %d", k_dtype);
  T2S_CHECK_ARG(!(exact_fp32 && k_dtype != T2S_F32), "ptr_scores: exact fp32 needs fp32 keys");
  dim3 grid((N + 63) / 64, B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (exact_fp32)
    hipLaunchKernelGGL(ptr_scores_f32_kernel, grid, block, 0, st, q, (const float*)k, mask, out, D, N, out_row_stride, col0, scale);
  else if (k_dtype == T2S_BF16)
    hipLaunchKernelGGL(ptr_scores_mfma_kernel<bf16_t>, grid, block, 0, st, q, (const bf16_t*)k, mask, out, D, N, out_row_stride, col0, scale);
  else
    hipLaunchKernelGGL(ptr_scores_mfma_kernel<float>, grid, block, 0, st, q, (const float*)k, mask, out, D, N, out_row_stride, col0, scale);
  T2S_CHECK_LAUNCH("ptr_scores");
  return 0;
}

extern "C" int t2s_question_pool(const float* qp, const float* w, const float* bias, const float* qmask, float* out, int B, int T,
                                 t2s_stream_t stream) {
  T2S_CHECK_ARG(qp && w && bias && qmask && out, "question_pool: null pointer");
  T2S_CHECK_ARG(B > 0 && T > 0 && T <= 64, "question_pool: bad shape B=%d T=%d (T <= 64)", B, T);
  hipLaunchKernelGGL(question_pool_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, qp, w, bias, qmask, out, T);
  T2S_CHECK_LAUNCH("question_pool");
  return 0;
}

extern "C" int t2s_attention_score(const float* q, const void* k, int64_t k_batch_stride, const float* mask, float* score, int B, int M,
                                   int k_dtype, t2s_stream_t stream) {
  T2S_CHECK_ARG(q && k && mask && score, "attention_score: null pointer");
  T2S_CHECK_ARG(B > 0 && M > 0 && k_batch_stride >= (int64_t)M * T2S_HIDDEN && k_batch_stride % 4 == 0, "attention_score: bad shape");
  T2S_CHECK_ARG(k_dtype == T2S_F32 || k_dtype == T2S_BF16, "attention_score: bad dtype %d", k_dtype);
  const int64_t rows = (int64_t)B * M;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (k_dtype == T2S_BF16) hipLaunchKernelGGL(score_dot_kernel<bf16_t>, grid, block, 0, st, q, (const bf16_t*)k, k_batch_stride, score, M, rows);
  else hipLaunchKernelGGL(score_dot_kernel<float>, grid, block, 0, st, q, (const float*)k, k_batch_stride, score, M, rows);
  hipLaunchKernelGGL(score_softmax_kernel, dim3(B), block, 0, st, score, mask, M);
  T2S_CHECK_LAUNCH("attention_score");
  return 0;
}

extern "C" int t2s_ground_select(const float* frame_score, const float* frame_mask, const float* expo_frame, const int64_t* frame_id,
                                 const float* q_global, const void* ocr_feat, int64_t ocr_batch_stride, int ocr_dtype, const float* expo_ocr,
                                 const int64_t* temporal_id, const float* bbox, float* pos_obj_mask, float* neg_obj_mask,
                                 int64_t* ground_frame, float* new_ocr_mask, float* ocr_score, float* pos_ocr_mask,
                                 float* neg_ocr_mask, float* ground_box, int B, int F, int P, int frame_topk, int ocr_topk,
                                 t2s_stream_t stream) {
  T2S_CHECK_ARG(frame_score && frame_mask && expo_frame && frame_id && q_global && ocr_feat && expo_ocr && temporal_id && bbox &&
                    pos_obj_mask && neg_obj_mask && ground_frame && new_ocr_mask && ocr_score && pos_ocr_mask && neg_ocr_mask && ground_box,
                "ground_select: null pointer");
  T2S_CHECK_ARG(B > 0 && B <= 65535 && F > 0 && P > 0 && F <= 4096 && P <= 4096, "ground_select: bad shape B=%d F=%d P=%d", B, F, P);
  T2S_CHECK_ARG(frame_topk > 0 && frame_topk <= F && ocr_topk > 0 && ocr_topk <= P, "ground_select: top-k larger than the candidate set");
  hipStream_t st = (hipStream_t)stream;
  const int N = F * P;
  hipLaunchKernelGGL(select_frames_kernel, dim3(B), dim3(64), 4 * F * sizeof(float), st, frame_score, frame_mask, expo_frame, frame_id,
                     pos_obj_mask, neg_obj_mask, ground_frame, F, frame_topk);
  hipLaunchKernelGGL(new_ocr_mask_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, temporal_id, ground_frame, new_ocr_mask, N, frame_topk);
  if (int e = t2s_attention_score(q_global, ocr_feat, ocr_batch_stride, new_ocr_mask, ocr_score, B, N, ocr_dtype, stream)) return e;
  hipLaunchKernelGGL(select_ocr_kernel, dim3(F, B), dim3(64), 4 * P * sizeof(float), st, ocr_score, new_ocr_mask, expo_ocr, bbox,
                     pos_ocr_mask, neg_ocr_mask, ground_box, F, P, ocr_topk);
  T2S_CHECK_LAUNCH("ground_select");
  return 0;
}
