// standalone check + timing of the ping-pong forward kernel against attn_fwd_bf16_kernel (same inputs)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "attn_common.h"
hipError_t launch_attn_fwd_pp_bf16(const AttnParams& p, hipStream_t st);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, L = argc > 2 ? atoi(argv[2]) : 10132, useidx = argc > 3 ? atoi(argv[3]) : 0;
  const float amp = argc > 4 ? atof(argv[4]) : 3.4f;
  const int H = 12, ND = 12;
  const size_t nqkv = (size_t)B * L * 3 * 768, no = (size_t)B * L * 768, nl = (size_t)B * H * L;
  std::vector<uint16_t> h(nqkv);
  srand(1);
  for (size_t i = 0; i < nqkv; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f) * amp; uint32_t u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  void *qkv, *out0, *out1; float *lse0, *lse1; int32_t *idx = nullptr, *cnt = nullptr;
  CK(hipMalloc(&qkv, nqkv * 2)); CK(hipMalloc(&out0, no * 2)); CK(hipMalloc(&out1, no * 2));
  CK(hipMalloc(&lse0, nl * 4)); CK(hipMalloc(&lse1, nl * 4));
  CK(hipMemcpy(qkv, h.data(), nqkv * 2, hipMemcpyHostToDevice));
  AttnParams p{};
  p.q = qkv; p.k = (char*)qkv + 768 * 2; p.v = (char*)qkv + 2 * 768 * 2;
  p.B = B; p.H = H; p.Lq = L; p.idx_cap = L; p.n_dec = ND; p.dec_q0 = L - ND;
  p.q_rs = 3 * 768; p.q_bs = (int64_t)L * 3 * 768; p.kv_rs = 3 * 768; p.kv_bs = p.q_bs; p.o_rs = 768; p.o_bs = (int64_t)L * 768;
  p.scale = 0.125f; p.drop_thresh = 0; p.drop_inv = 1.f;
  if (useidx) {   // keep ~70 % of the prefix keys, ragged per sample
    std::vector<int32_t> hi((size_t)B * L), hc(B);
    for (int b = 0; b < B; ++b) {
      int n = 0;
      for (int j = 0; j < L - ND; ++j) if ((rand() % 100) < 70 - 3 * (b % 4)) hi[(size_t)b * L + n++] = j;
      hc[b] = n;
      for (int j = 0; j < ND; ++j) hi[(size_t)b * L + n + j] = L - ND + j;
    }
    CK(hipMalloc(&idx, hi.size() * 4)); CK(hipMalloc(&cnt, B * 4));
    CK(hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(cnt, hc.data(), B * 4, hipMemcpyHostToDevice));
    p.kv_idx = idx; p.kv_cnt = cnt;
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms0, ms1;
  p.out = out0; p.lse = lse0;
  launch_attn_fwd_bf16(p, 0); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0)); for (int i = 0; i < 5; ++i) launch_attn_fwd_bf16(p, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms0, e0, e1));
  p.out = out1; p.lse = lse1;
  CK(launch_attn_fwd_pp_bf16(p, 0)); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0)); for (int i = 0; i < 5; ++i) CK(launch_attn_fwd_pp_bf16(p, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms1, e0, e1));
  std::vector<uint16_t> o0(no), o1(no); std::vector<float> l0(nl), l1(nl);
  CK(hipMemcpy(o0.data(), out0, no * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), out1, no * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(l0.data(), lse0, nl * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(l1.data(), lse1, nl * 4, hipMemcpyDeviceToHost));
  double md = 0, ml = 0, mo = 0; size_t nan1 = 0;
  for (size_t i = 0; i < no; ++i) { double a = bf2f(o0[i]), c = bf2f(o1[i]); if (!(c == c)) ++nan1; md = fmax(md, fabs(a - c)); mo = fmax(mo, fabs(a)); }
  for (size_t i = 0; i < nl; ++i) { if (!(l1[i] == l1[i])) ++nan1; else ml = fmax(ml, fabs((double)l0[i] - l1[i])); }
  const double fl = 2.0 * 2.0 * B * H * (double)L * L * 64;
  printf("B=%d L=%d idx=%d amp=%.1f | old %.3f ms %.0f TF/s | pp %.3f ms %.0f TF/s | max|dO| %.3g (max|O| %.3g) max|dLSE| %.3g nan/poison %zu\n", B, L, useidx, amp,
         ms0 / 5, fl / (ms0 / 5 * 1e-3) / 1e12, ms1 / 5, fl / (ms1 / 5 * 1e-3) / 1e12, md, mo, ml, nan1);
  return 0;
}
