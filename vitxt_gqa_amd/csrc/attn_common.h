// Shared pieces of the flash-style attention kernels (forward, dK/dV, dQ) for head_dim 64.
#pragma once
#include "common.h"

struct AttnParams {
  const void* q;
  const void* k;
  const void* v;
  const void* o;        // forward output / saved output (backward)
  const void* dout;     // backward
  void* out;            // forward
  void* dq;
  void* dk;
  void* dv;
  float* lse;           // forward: written; backward: read
  float* delta;         // backward: rowsum(dO * O)
  const int32_t* kv_idx;
  const int32_t* kv_cnt;
  int B, H, Lq, idx_cap, n_dec, dec_q0;
  int64_t q_rs, q_bs, kv_rs, kv_bs, o_rs, o_bs;
  float scale;
  // attention-probability dropout (BertSelfAttention: dropout(softmax(.)) before .V); thresh == 0 disables it
  const uint32_t* drop_rowkey;   // [B, H, ceil(Lq/2)] per-(sample, head, query pair) hash keys (attn_drop_rowkeys_kernel)
  uint32_t drop_thresh;          // drop iff byte < thresh, thresh = round(p * 256)
  float drop_inv;                // 1 / (1 - thresh/256)
};

// ---- attention-probability dropout.  keep(q, kpos) is a stateless function of (seed, sample, head, q, kpos) so the
// forward, dQ and dK/dV kernels regenerate the same mask in their different register layouts; kpos is the POSITION in
// the compacted key list.  One 32-bit word serves a 2x2 block {q, q^1} x {kpos, kpos^1} (a byte per element), which is
// the largest block both layouts share: with the query on the lane a register pair holds two consecutive keys, with
// the key on the lane two consecutive queries.  rowkey[b,h,q>>1] is a full-quality hash (precomputed table); the
// per-block mixer uses only 24-bit multiplies, shifts and xors (v_mul_lo_u32 is quarter rate on CDNA).
__device__ __forceinline__ uint32_t attn_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t attn_drop_block(uint32_t rowkey, uint32_t kp2) {
  uint32_t x = rowkey + __umul24(kp2, 0x9E3779u);
  x ^= x >> 15;
  x = __umul24(x, 0x2C1B3Du) + (x >> 9);
  x ^= x >> 13;
  x = __umul24(x, 0x6D2B79u) ^ (x >> 11);
  return x;
}
// generic per-element form (fp32 kernels, mask export): byte (q&1)*2 + (kpos&1) of the block word
__device__ __forceinline__ bool attn_drop_keep(uint32_t rowkey, int q, int kpos, uint32_t thresh) {
  const uint32_t x = attn_drop_block(rowkey, (uint32_t)kpos >> 1);
  return ((x >> (8 * ((q & 1) * 2 + (kpos & 1)))) & 0xFFu) >= thresh;
}
// packed form: returns the AND-mask for a bf16x2 word holding the block's elements whose bytes are `sel`-selected
// (sel = v_perm selector placing the two bytes in the low byte of each 16-bit half): 0xFFFF per kept half.
__device__ __forceinline__ uint32_t attn_drop_pair_mask(uint32_t x, uint32_t sel, uint32_t thresh2 /* thresh | thresh << 16 */) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const uint32_t t = __builtin_amdgcn_perm(0u, x, sel);
  const s16x2 d = __builtin_bit_cast(s16x2, t) - __builtin_bit_cast(s16x2, thresh2);
  const s16x2 m = d >> 15;                       // 0xFFFF where byte < thresh (dropped)
  return ~__builtin_bit_cast(uint32_t, m);
}
__device__ __forceinline__ uint32_t attn_drop_sel(int b0, int b1) { return 0x0c000c00u | ((uint32_t)b1 << 16) | (uint32_t)b0; }

void launch_attn_drop_rowkeys(uint32_t* rowkey, int B, int H, int Lq, uint64_t seed, hipStream_t st);
int attn_setup_dropout(AttnParams& p, float drop_p, uint64_t drop_seed, uint32_t* drop_ws, hipStream_t st, const char* who);

// ---- LDS tile image shared by every bf16 tile (K, V, Q, dO): rows of 64 bf16 = 128 B = eight
// 16-byte chunks, chunk c of row r stored at chunk position c ^ f(r).  f is chosen so that BOTH
// read kinds used on a tile are bank-conflict free on gfx950 (MI355X_MICROARCH.md, LDS):
//   * ds_read_b128 row reads (MFMA 32x32x16 A operand: lane = row, 16-lane service groups
//     {0-3,12-15,20-27} / {4-11,16-19,28-31}): same-parity rows of a group get 8 distinct f;
//   * ds_read_b64_tr_b16 transposed reads (4 consecutive rows x 64 B per 32-lane half): rows
//     r, r+1 differ in the 128-B half of the 256-B bank row, rows r+2, r+3 land in the other
//     64-B segment (bit 2 of f = bit 1 of r).
__device__ __forceinline__ int tile_f(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ tile_f(row)) << 4); }

// A/B operand of v_mfma_f32_32x32x16_bf16 read as a row fragment: lane (r = lane&31, h = lane>>5)
// gets elements [row0 + r][16*s + 8*h .. +7].
__device__ __forceinline__ bf16x8 lds_row_frag(const char* tile, int row, int s, int lh) {
  return *reinterpret_cast<const bf16x8*>(tile + tile_off(row, 2 * s + lh));
}

// Transposed fragment (rows of the tile are the MFMA K dimension, columns its M dimension):
// returns for lane (r = lane&31 -> column dblock*32 + r, h = lane>>5) the 8 elements of tile rows
//   rbase + 8*(j>>2) + 4*h + (j&3),  j = 0..7
// which is exactly the k order of an accumulator tile (registers 8s..8s+7) used as the other
// operand (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand").
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* tile, int rbase, int dblock, int lane) {
  const int lh = lane >> 5, g1 = (lane >> 4) & 1, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
  const int chunk = 4 * dblock + 2 * g1 + (pp >> 1);
  const int r0 = rbase + 4 * lh + qq;
  const int r1 = r0 + 8;
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + tile_off(r0, chunk) + ((pp & 1) << 3)));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + tile_off(r1, chunk) + ((pp & 1) << 3)));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 c = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, c);
}

// accumulator registers 8*s .. 8*s+7 -> bf16 operand fragment
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16_t)a[8 * s + j];
  return r;
}

// row index (0..31) inside a 32x32 accumulator tile held in register `reg` by lane-half `lh`
__device__ __forceinline__ int acc_row(int reg, int lh) { return (reg & 3) + 8 * (reg >> 2) + 4 * lh; }

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// bf16 forward kernel lives in attn_fwd_bf16.hip
void launch_attn_fwd_bf16(const AttnParams& p, hipStream_t st);
// bf16 dK/dV kernel lives in attn_dkdv_bf16.hip
void launch_attn_dkdv_bf16(const AttnParams& p, int max_keys, hipStream_t st);

#define LOG2E 1.4426950408889634f

// raw v_exp_f32 (one instruction; results below 2^-126 flush to 0, which is what a softmax wants)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
