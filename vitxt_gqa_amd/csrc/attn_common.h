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
  int kblocks;          // dK/dV: 128-key blocks per (sample, head) in the launch (static bound on the visible keys)
  int64_t q_rs, q_bs, kv_rs, kv_bs, o_rs, o_bs;
  float scale;
  // attention-probability dropout (BertSelfAttention: dropout(softmax(.)) before .V); thresh == 0 disables it
  uint32_t drop_seed_lo, drop_seed_hi;
  uint32_t drop_thresh;          // 16-bit threshold round(p * 65536) in [1, 65535]: drop iff the score's 16-bit value is below it
  float drop_inv;                // 1 / (1 - thresh / 65536)
  // backward, optional: row_valid [B, valid_len] bytes, 0 = the prefix row is NOT in the key list.  The dQ kernel, which
  // visits every (row, head) anyway, then writes the zeros of that row's dK / dV slices, so the caller need not zero-fill
  // the gradient buffer (of the rows >= valid_len only this call's decoder rows [dec_q0, dec_q0 + n_dec) are listed; in the
  // shared-prefix layout of the three MMT passes the decoder rows of the other two passes follow the prefix as well)
  const uint8_t* row_valid;
  int valid_len;
};

// ---- attention-probability dropout.  keep(q, kpos) is a stateless function of (seed, sample, head, q, kpos) so that
// the forward, dQ and dK/dV kernels regenerate the same mask in their different register layouts; kpos is the POSITION
// in the compacted key list.  The function is built to cost ~2.5 VALU instructions per score in kernels that are VALU
// bound:   t(q, kpos) = (rowkey16(q) * colkey16(kpos) mod 2^16) read as int16,   keep iff t >= thresh - 32768   (thresh = round(65536 p))
// where rowkey16 / colkey16 are ODD 16-bit values cut from full-quality 32-bit hashes of (seed, sample, head, q) and (seed, sample,
// head, kpos).  Those hashes are per-lane constants on the stationary axis and are computed once per tile (32 threads, one
// key pair each, staged in LDS) on the streamed axis, so the per-score work is multiply + compare - and it is
// done on TWO scores per instruction with packed 16-bit math: a bf16x2 word of P holds two consecutive keys of the
// lane's query (forward, dQ) or two consecutive queries of the lane's key (dK/dV), exactly the two 16-bit halves.
// Multiplication by an odd number is a bijection of the odd residues mod 2^16, so for a fixed row the values t are
// independent and uniform over keys (colkey16 independent uniform odd values) and vice versa.  NOT the reference's i.i.d.
// Philox draw: two rows are related by t2 = (rk2 / rk1) t1, a multiplicative scramble; rows share a mask only if their keys
// collide (2^-15 per pair) and are visibly correlated only for a handful of special key ratios (1, -1, small odd numbers and
// their inverses).
// Round 4: the ROW key is a function of (row, key WINDOW): a window = ATTN_DROP_KWIN = 384 consecutive key-list positions, and
// rowkey16w(h, w) = high 16 bits of h * (odd 32-bit hash of w) | 1 with h the row's 32-bit hash.  With ONE 16-bit key per row, 1 100 -
// 1 600 of the 51 M row pairs of a (sample, head) at L = 10 132 drew the same key and with it the same mask over ALL keys (a quarter
// of the rows had such a twin: tests/test_dropout_gpu.py).  Now two rows collide per window (independently, 2^-15 each): a pair
// shares its mask over 384 keys of 10 132 at most, never everywhere.  Cost: none in the key-stationary kernels (a workgroup's keys lie
// in one window: 384 is the fused kernel's block, three of the dK/dV kernel's), one multiply + shift + or per 64-key tile and query
// block in the query-stationary ones.  (The first form of this function, (rowkey ^ colkey) * 0x9E37, cost one instruction more per pair and left
// 0.8 % of the row pairs correlated beyond 6.5 sigma: tests/test_dropout_gpu.py measures both properties.)
__device__ __forceinline__ uint32_t attn_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t attn_drop_salt(uint32_t seed_lo, uint32_t seed_hi, uint32_t bh) {
  return attn_hash32(seed_lo ^ attn_hash32(seed_hi ^ (bh * 0x9E3779B1u)));
}
__device__ __forceinline__ uint32_t attn_drop_rowhash(uint32_t salt, int q) { return attn_hash32(salt + (uint32_t)q * 0x85EBCA6Bu); }
// (the window multiplier is itself a HASH of the window index, a wave-uniform scalar computation: with small odd multiples
//  (2 w + 1) c of one constant, two rows whose hashes differ by d with d c mod 2^32 small collided in EVERY window at once - 4 identical
//  row pairs at L = 2 048 in the first form of this function, against 0 expected)
__device__ __forceinline__ uint32_t attn_drop_rowkey16w(uint32_t rowhash, int kwin) {
  return ((rowhash * (attn_hash32((uint32_t)kwin * 0x9E3779B1u + 0x5bd1e995u) | 1u)) >> 16) | 1u;
}
// ... and, symmetrically, the COLUMN key is per (key-list position, ROW window of ATTN_DROP_QWIN = 256 query rows): two keys whose
// column keys collide are dropped together for 256 queries, not for every query of the head.  Free in the query-stationary kernels
// (a workgroup's 128 or 256 query rows lie in one window: the staged keys are hashed with it), a re-hash of the lane's keys every
// fourth 64-row tile in the key-stationary ones.
__device__ __forceinline__ uint32_t attn_drop_colkey16(uint32_t salt, int kpos, int qwin) {
  const uint32_t ch = attn_hash32((salt ^ 0xC2B2AE35u) + (uint32_t)kpos * 0x27D4EB2Fu);
  return ((ch * (attn_hash32((uint32_t)qwin * 0x85EBCA6Bu + 0x1b873593u) | 1u)) >> 16) | 1u;
}
constexpr int ATTN_DROP_KWIN = 384;          // key-list positions per row-key window (the fused backward's key block)
constexpr int ATTN_DROP_QWIN = 256;          // query rows per column-key window (the forward's workgroup)
__device__ __forceinline__ uint32_t attn_drop_rowkey16(uint32_t salt, int q, int kwin) { return attn_drop_rowkey16w(attn_drop_rowhash(salt, q), kwin); }
// generic per-element form (fp32 kernels, mask export).  The 16-bit product is read as a SIGNED number and compared with
// thresh - 32768: the same drop probability thresh / 65536 as an unsigned 16-bit compare, without the flip of the top
// bit the unsigned form needs before a signed saturating subtract (one VALU instruction per score pair in VALU-bound loops).
__device__ __forceinline__ bool attn_drop_keep16(uint32_t rk16, uint32_t ck16, uint32_t thresh) {
  const int t = (int)(short)((rk16 * ck16) & 0xFFFFu);
  return t >= (int)thresh - 32768;
}
// packed form: a2 / b2 hold the row and column keys of two scores in their 16-bit halves; returns per half a signed 16-bit
// value that is NEGATIVE iff the score is dropped: v_pk_mul_lo_u16, v_pk_sub_i16 clamp - two instructions for two scores.
__device__ __forceinline__ uint32_t attn_drop_pair_diff(uint32_t a2, uint32_t b2, uint32_t thresh2s /* attn_drop_thresh2s(thresh) */) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const u16x2 t = __builtin_bit_cast(u16x2, a2) * __builtin_bit_cast(u16x2, b2);
  const s16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, t), __builtin_bit_cast(s16x2, thresh2s));
  return __builtin_bit_cast(uint32_t, d);
}
// thresh - 32768 as a 16-bit pattern, in both halves
__device__ __forceinline__ uint32_t attn_drop_thresh2s(uint32_t thresh) { return ((thresh ^ 0x8000u) & 0xFFFFu) * 0x10001u; }
// 0xFFFF in every DROPPED half (v_pk_ashrrev_i16), for clearing halves of packed bf16 words
__device__ __forceinline__ uint32_t attn_drop_pair_dropped(uint32_t a2, uint32_t b2, uint32_t thresh2s) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 m = __builtin_bit_cast(s16x2, attn_drop_pair_diff(a2, b2, thresh2s)) >> 15;
  return __builtin_bit_cast(uint32_t, m);
}
// The KEEP form of the same test: 0xFFFF in every KEPT half (kept iff t >= ths  <=>  (ths - 1) - t < 0: v_pk_mul_lo_u16,
// v_pk_sub_i16 clamp with the operands the other way round, v_pk_ashrrev_i16).  Such a word clears a packed bf16 pair with ONE v_and_b32
// and an fp32 value with ONE v_and_b32_sdwa (attn_drop_keep_lo / _hi below: the 16-bit half, sign-extended by the operand selector, is
// the 32-bit mask) - no per-score mask expansion.  thresh2k = attn_drop_thresh2k(thresh).
__device__ __forceinline__ uint32_t attn_drop_thresh2k(uint32_t thresh) { return (((thresh ^ 0x8000u) - 1u) & 0xFFFFu) * 0x10001u; }
__device__ __forceinline__ uint32_t attn_drop_pair_kept(uint32_t a2, uint32_t b2, uint32_t thresh2k) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const u16x2 t = __builtin_bit_cast(u16x2, a2) * __builtin_bit_cast(u16x2, b2);
  const s16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, thresh2k), __builtin_bit_cast(s16x2, t));
  return __builtin_bit_cast(uint32_t, d >> 15);
}
// the three instructions of attn_drop_pair_kept one by one, for kernels that software-pipeline them over consecutive score pairs
// (a dependent pair of packed 16-bit instructions back to back costs a wait state: cdna hazard "VALU write with op_sel -> VALU read")
__device__ __forceinline__ uint32_t attn_drop_kept_mul(uint32_t a2, uint32_t b2) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a2) * __builtin_bit_cast(u16x2, b2));
}
__device__ __forceinline__ uint32_t attn_drop_kept_sub(uint32_t t2, uint32_t thresh2k) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, thresh2k), __builtin_bit_cast(s16x2, t2)));
}
// (drop form of the middle stage: negative halves = dropped, thresh2s = attn_drop_thresh2s(thresh))
__device__ __forceinline__ uint32_t attn_drop_dropped_sub(uint32_t t2, uint32_t thresh2s) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, t2), __builtin_bit_cast(s16x2, thresh2s)));
}
__device__ __forceinline__ uint32_t attn_drop_kept_mask(uint32_t d2) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, d2) >> 15);
}
__device__ __forceinline__ float attn_drop_keep_lo(float x, uint32_t kword) {
  float r;
  asm("v_and_b32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(x), "v"(kword));
  return r;
}
__device__ __forceinline__ float attn_drop_keep_hi(float x, uint32_t kword) {
  float r;
  asm("v_and_b32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(x), "v"(kword));
  return r;
}
// word of two bf16 probabilities with the dropped halves cleared (v_bfi_b32)
__device__ __forceinline__ uint32_t attn_drop_apply(uint32_t w, uint32_t dropped) { return w & ~dropped; }

// per-score 32-bit masks (all ones = dropped) of the two halves, and "zero the float if dropped" (v_bfe_i32 / v_ashrrev_i32,
// v_bfi_b32): selects without compares
// (work on the diff word or on the 0xFFFF mask word alike: only the sign bit of each half is used)
__device__ __forceinline__ uint32_t attn_drop_lo32(uint32_t m) { return (uint32_t)((int32_t)(m << 16) >> 31); }
__device__ __forceinline__ uint32_t attn_drop_hi32(uint32_t m) { return (uint32_t)((int32_t)m >> 31); }
__device__ __forceinline__ float attn_drop_zero(float x, uint32_t m32) {
  return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) & ~m32);
}

// The same two selects as single instructions each (v_bfe_i32 / v_ashrrev_i32 for the 32-bit mask, v_bfi_b32 to clear): written
// in C the compiler turns "x & ~sign_mask" into a compare + v_cndmask per score and an extra v_and per pair (5 instructions per
// score pair instead of 4 in the issue-bound phase of the fused backward)
__device__ __forceinline__ float attn_drop_zero_lo(float x, uint32_t mword) {
  uint32_t m32;
  float r;
  asm("v_bfe_i32 %0, %1, 0, 16" : "=v"(m32) : "v"(mword));
  asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(r) : "v"(m32), "v"(x));
  return r;
}
__device__ __forceinline__ float attn_drop_zero_hi(float x, uint32_t mword) {
  uint32_t m32;
  float r;
  asm("v_ashrrev_i32 %0, 16, %1" : "=v"(m32) : "v"(mword));
  asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(r) : "v"(m32), "v"(x));
  return r;
}

int attn_setup_dropout(AttnParams& p, float drop_p, uint64_t drop_seed, hipStream_t st, const char* who);

// ---- XCD-aware workgroup -> tile mapping.  MI355X hands consecutive workgroup ids round-robin to its 8 XCDs, each with
// its own 4 MB L2.  The workgroups that share operands - the query blocks of one (sample, head), which all stream the same
// K/V rows (forward, dQ), or its key blocks, which all stream the same Q/dO rows (dK/dV) - must therefore NOT have
// consecutive ids, or every XCD's L2 fetches every (sample, head)'s K/V from HBM (measured: 8.5 GB per launch against
// 4 GB algorithmic).  A 1-D grid is launched; workgroup id L runs on XCD L % 8 as its (L / 8)-th workgroup: that XCD walks
// the (sample, head) pairs  xcd, xcd + 8, xcd + 16, ...  block after block.  Taking every 8th pair (rather than a
// contiguous eighth of them) keeps the XCDs balanced when the samples of a batch differ in work - e.g. the stacked
// ref / pos / neg passes, whose visible key counts differ by two orders of magnitude.  Returns false for padding ids.
constexpr int T2S_XCDS = 8;
__device__ __forceinline__ bool attn_xcd_tile(int nblk, int H, int B, int& blk, int& h, int& b) {
  const int L = (int)blockIdx.x;
  const int slot = L / T2S_XCDS;
  const int g = slot / nblk;
  const int bh = g * T2S_XCDS + L % T2S_XCDS;
  if (bh >= H * B) return false;
  blk = slot - g * nblk;
  b = bh / H;
  h = bh - b * H;
  return true;
}
inline unsigned attn_xcd_grid(int nblk, int H, int B) {
  const long groups = ((long)H * B + T2S_XCDS - 1) / T2S_XCDS;
  return (unsigned)(groups * T2S_XCDS * nblk);
}

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
// fused 5-product bf16 backward (delta + housekeeping, main kernel, dQ cast) lives in attn_bwd_fused_bf16.hip
int launch_attn_bwd_fused_bf16(const AttnParams& p, int max_keys, void* workspace, size_t workspace_bytes, int handoff, hipStream_t st);
size_t attn_bwd_fused_workspace_bytes(int B, int H, int Lq);

#define LOG2E 1.4426950408889634f

// raw v_exp_f32 (one instruction; results below 2^-126 flush to 0, which is what a softmax wants)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
