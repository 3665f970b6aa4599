// Global-norm gradient clipping + Adam as three multi-tensor passes for gfx950 (HBM-bound):
//   BaseTrainer._backward (pythia/trainers/base_trainer.py:262-272): clip_gradients (pythia/utils/general.py:32-41:
//   torch.nn.utils.clip_grad_norm_(params, 0.25)) followed by optimizer.step() of torch.optim.Adam (build_utils.py:54-83,
//   configs/t2s_abinet.yml: lr 1e-4, eps 1e-8, weight_decay 0).
// The parameters stay separate tensors (the reference's state_dict / optimizer state layout is kept); a descriptor table in
// device memory - one row (param, grad, exp_avg, exp_avg_sq, numel) per tensor - and a chunk table (tensor, chunk-in-tensor)
// let one launch walk all of them: 64 Ki elements per workgroup, 16-byte loads and stores.
//   pass 1  t2s_grad_sqnorm   sum of squares of every gradient chunk            (reads 4 B / element)
//   pass 2  t2s_clip_coef     total norm (fp64 sum of the partials) and the clip coefficient min(1, max_norm / (norm + 1e-6))
//   pass 3  t2s_adam_step     g *= coef (written back: the reference clips p.grad in place), Adam moments, parameter update
//                             (reads 16 B, writes 16 B per element)
#include "common.h"

namespace {

constexpr int OPT_CHUNK = 65536;      // elements per workgroup

struct AdamArgs {
  float lr[8];                        // learning rate of each param group (<= 8 groups)
  float beta1, beta2, eps, bc1, bc2_sqrt;      // bias corrections 1 - beta^step and sqrt(1 - beta2^step)
  int write_grad;
};

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const int64_t* __restrict__ desc, const int32_t* __restrict__ chunks,
                                                          float* __restrict__ partials) {
  __shared__ float sh[4];
  const int t = chunks[2 * blockIdx.x], c = chunks[2 * blockIdx.x + 1];
  const float* __restrict__ g = reinterpret_cast<const float*>(desc[5 * t + 1]);
  const int64_t n = desc[5 * t + 4];
  const int64_t lo = (int64_t)c * OPT_CHUNK, hi = lo + OPT_CHUNK < n ? lo + OPT_CHUNK : n;
  float s = 0.f;
  const bool vec = (reinterpret_cast<uintptr_t>(g) & 15) == 0;
  if (vec) {
    int64_t i = lo + (int64_t)threadIdx.x * 4;
    for (; i + 3 < hi; i += 1024) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (i < hi)       // at most one thread holds the ragged tail of the tensor
      for (; i < hi; ++i) s += g[i] * g[i];
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += g[i] * g[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partials, int n, float max_norm, float* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(sh[0]);
    const float coef = max_norm / (norm + 1e-6f);          // torch.nn.utils.clip_grad_norm_: clamped to 1
    out[0] = norm;
    out[1] = coef < 1.f ? coef : 1.f;
  }
}

// ---- the sticky status of a train step (a hand-off wait of the fused attention backward timed out: its dQ rows are NaN by
// construction).  status_accumulate ORs the status word of a launch's workspace into a word that outlives the workspace; status_gate
// turns the step into a no-op (clip coefficient -1: adam_step_kernel returns at once, parameters, moments and gradients untouched)
// and the reported norm into NaN; the host raises when it reads the word (optim.py).
__global__ void status_accumulate_kernel(const unsigned* __restrict__ word, unsigned* __restrict__ sticky) {
  const unsigned v = *word;
  if (v) atomicOr(sticky, v);
  atomicAdd(sticky + 1, 1u);          // launches seen
  // the two status bits once more as 0 / 1 FLAGS of their own (words 2, 3): a MAX reduction over ranks - the one collective
  // GradBuckets.finish() spends on the word - is an OR for flags, while the MAX of two OR-ed words (1 on one rank, 2 on another) loses a bit
  if (v & 1u) atomicOr(sticky + 2, 1u);
  if (v & 2u) atomicOr(sticky + 3, 1u);
}
__global__ void status_gate_kernel(const unsigned* __restrict__ sticky, float* __restrict__ norm_coef) {
  if ((sticky[0] | sticky[2] | sticky[3]) != 0u) {
    norm_coef[0] = __builtin_nanf("");
    norm_coef[1] = -1.f;
  }
}

__device__ __forceinline__ void adam_elem(float& p, float& g, float& m, float& v, float coef, float lr, const AdamArgs& a) {
  g *= coef;
  m = m + (g - m) * (1.f - a.beta1);                        // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.beta2 + (1.f - a.beta2) * g * g;                // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p -= (lr / a.bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(const int64_t* __restrict__ desc, const int32_t* __restrict__ chunks,
                                                        const int32_t* __restrict__ group_of, const float* __restrict__ coef_p, AdamArgs a) {
  const int t = chunks[2 * blockIdx.x], c = chunks[2 * blockIdx.x + 1];
  float* __restrict__ p = reinterpret_cast<float*>(desc[5 * t]);
  float* __restrict__ g = reinterpret_cast<float*>(desc[5 * t + 1]);
  float* __restrict__ m = reinterpret_cast<float*>(desc[5 * t + 2]);
  float* __restrict__ v = reinterpret_cast<float*>(desc[5 * t + 3]);
  const int64_t n = desc[5 * t + 4];
  const float lr = a.lr[group_of[t]];
  const float coef = coef_p ? coef_p[1] : 1.f;
  if (coef < 0.f) return;          // the step was gated off (status_gate_kernel): nothing is touched
  const int64_t lo = (int64_t)c * OPT_CHUNK, hi = lo + OPT_CHUNK < n ? lo + OPT_CHUNK : n;
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  if (vec) {
    int64_t i = lo + (int64_t)threadIdx.x * 4;
    for (; i + 3 < hi; i += 1024) {
      f32x4 pv = *reinterpret_cast<const f32x4*>(p + i), gv = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mv = *reinterpret_cast<const f32x4*>(m + i), vv = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float p1 = pv[j], g1 = gv[j], m1 = mv[j], v1 = vv[j];
        adam_elem(p1, g1, m1, v1, coef, lr, a);
        pv[j] = p1; gv[j] = g1; mv[j] = m1; vv[j] = v1;
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m + i) = mv;
      *reinterpret_cast<f32x4*>(v + i) = vv;
      if (a.write_grad) *reinterpret_cast<f32x4*>(g + i) = gv;
    }
    if (i < hi)
      for (; i < hi; ++i) {
        float pv = p[i], gv = g[i], mv = m[i], vv = v[i];
        adam_elem(pv, gv, mv, vv, coef, lr, a);
        p[i] = pv; m[i] = mv; v[i] = vv;
        if (a.write_grad) g[i] = gv;
      }
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
      float pv = p[i], gv = g[i], mv = m[i], vv = v[i];
      adam_elem(pv, gv, mv, vv, coef, lr, a);
      p[i] = pv; m[i] = mv; v[i] = vv;
      if (a.write_grad) g[i] = gv;
    }
  }
}

}  // namespace

extern "C" int t2s_optim_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int t2s_status_accumulate(const void* workspace, int word, void* sticky, t2s_stream_t stream) {
  T2S_CHECK_ARG(workspace && sticky && word >= 0, "status_accumulate: bad arguments");
  hipLaunchKernelGGL(status_accumulate_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<const unsigned*>(workspace) + word,
                     reinterpret_cast<unsigned*>(sticky));
  T2S_CHECK_LAUNCH("status_accumulate");
  return 0;
}

extern "C" int t2s_status_gate(const void* sticky, float* norm_coef, t2s_stream_t stream) {
  T2S_CHECK_ARG(sticky && norm_coef, "status_gate: null pointer");
  hipLaunchKernelGGL(status_gate_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<const unsigned*>(sticky), norm_coef);
  T2S_CHECK_LAUNCH("status_gate");
  return 0;
}

extern "C" int t2s_grad_sqnorm(const int64_t* desc, const int32_t* chunks, int n_chunks, float* partials, t2s_stream_t stream) {
  T2S_CHECK_ARG(desc && chunks && partials && n_chunks > 0, "grad_sqnorm: null pointer / no chunks");
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, desc, chunks, partials);
  T2S_CHECK_LAUNCH("grad_sqnorm");
  return 0;
}

extern "C" int t2s_clip_coef(const float* partials, int n_chunks, float max_norm, float* norm_coef, t2s_stream_t stream) {
  T2S_CHECK_ARG(partials && norm_coef && n_chunks > 0 && max_norm > 0.f, "clip_coef: bad arguments");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, n_chunks, max_norm, norm_coef);
  T2S_CHECK_LAUNCH("clip_coef");
  return 0;
}

extern "C" int t2s_adam_step(const int64_t* desc, const int32_t* chunks, int n_chunks, const int32_t* group_of, const float* group_lr,
                             int n_groups, float beta1, float beta2, float eps, int step, const float* norm_coef, int write_grad,
                             t2s_stream_t stream) {
  T2S_CHECK_ARG(desc && chunks && group_of && group_lr && n_chunks > 0, "adam_step: null pointer / no chunks");
  T2S_CHECK_ARG(n_groups > 0 && n_groups <= 8, "adam_step: %d param groups (at most 8)", n_groups);
  T2S_CHECK_ARG(step > 0 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f, "adam_step: bad hyper-parameters");
  AdamArgs a = {};
  for (int i = 0; i < n_groups; ++i) a.lr[i] = group_lr[i];
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.write_grad = write_grad;
  hipLaunchKernelGGL(adam_step_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, desc, chunks, group_of, norm_coef, a);
  T2S_CHECK_LAUNCH("adam_step");
  return 0;
}
