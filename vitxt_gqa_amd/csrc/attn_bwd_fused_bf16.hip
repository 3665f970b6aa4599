// Fused bf16 flash-attention backward for gfx950, head_dim 64: FIVE matrix products per (query, key) pair.
//
// The two-kernel form (attn_dkdv_bf16.hip + attn_dq_bf16_kernel) needs no cross-workgroup sum but computes S = Q K^T and
// dP = dO V^T twice: 7 products.  Here ONE key-stationary kernel computes S and dP once and also forms dQ += dS K; what it
// costs is a sum of dQ across the key blocks of a (sample, head) - by an ordered hand-off of running sums (round 4, the shipped
// form) or with fp32 atomics (rounds 2-3, dq_mode 0): "dQ across the key blocks" below.  The atomic form sized the geometry
// (cdna_hip_programming.md Guideline 12, MI355X_MICROARCH.md "Global float atomics": ~1.3 TB/s chip-wide):
// a workgroup of KB keys adds a [q-tile x 64] fp32 tile per query tile = 256 B per query row per 10*64*KB FLOPs, i.e.
// 2.5 * KB FLOP per atomic byte.  KB = 128 (the dK/dV kernel's block) caps the kernel at 0.42 PFLOP/s, below what the
// two-kernel form already reaches; KB = 384 puts the cap at 1.25 PFLOP/s.  So: 4 waves, ONE wave per SIMD with the whole
// 512-register file, 96 keys per wave:
//   registers  dK^T / dV^T of 96 keys (192 accumulators), V fragments (48), the query tile's A operands (64: Q, dO rows and
//              Q^T, dO^T transposed fragments, shared by the wave's three key blocks), row constants (32), S / dP (32)
//   LDS        K image of the workgroup's 384 keys (48 KB, pre-scaled by scale*log2e: B operand of S by row reads, B operand
//              of dQ by transposed reads), dS^T image [key][query] of the current query tile (48 KB), double-buffered Q / dO
//              tiles + row constants (33 KB)
// Per 64-row query tile:  phase A, per wave and 32-row sub-block: S, dP, P = exp2(S'), dS = P dP' (row constants seeded through
// the MFMA C operand), dV^T += dO^T P, dK^T += Q^T dS with P / dS straight from the accumulators (key on the lane), dS
// also stored transposed into the dS^T image (8 bytes per lane per 4 registers);  barrier;  phase B: wave w owns the
// 32 x 32 tile (query sub-block w >> 1, dim block w & 1) of dQ = dS K over ALL 384 keys (both operands by
// ds_read_b64_tr_b16 from the two images) and adds it to the running sum of the pair's earlier key blocks (hand-off) or to the fp32
// dQ buffer (atomics: each accumulator register is two 128-byte row segments, the shape the atomics run at full rate for).
// In the straight-line sweep (all 384 keys valid prefix keys: almost every workgroup of a long list) phase B of tile t - 1 runs
// INSIDE slots 0 and 1 of phase A of tile t ("ILV" below, shipped since round 5); the edge sweep and the tail launch keep it serial.
#include <stdlib.h>

#include <type_traits>

#include "attn_common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifndef FB_ABL_NOBAR
#define FB_ABL_NOBAR 0                            // TIMING-ONLY ablation (results wrong, tools/ablate): 1 no barrier behind slot 1, 2 no barrier at the end of the tile
#endif
// FB_SOFT_SYNC: in the interleaved sweep the dS^T image's readers (the dQ steps of slots 0 / 1) are retired ahead of the tile's first dS^T
// store by an LDS COUNTER instead of the workgroup barrier behind slot 1: every wave adds 1 when its slot 1 is through, and a wave waits
// for 4 (tile + 1) only where it needs it - in front of its first dS^T store, which moves from inside slot 2 to just behind it.  The
// barrier made every wave wait for the slowest one's slot 1 before its own slot 2; the counter gives the waves a slot of slack
// (removing the barrier outright, results wrong, measured -2.2 %: profiles/r06_fb_ablation_stamps.txt).
#ifndef FB_SOFT_SYNC
#define FB_SOFT_SYNC 0
#endif
#ifndef FB_DQ_DEPTH
#define FB_DQ_DEPTH 3                             // 16-key steps the operand reads of the interleaved dQ product run ahead of their MFMA (see afA / bfA)
#endif
constexpr int FB_KB = 3;                          // 32-key blocks per wave
constexpr int FB_WKEYS = 32 * FB_KB;              // 96 keys per wave
constexpr int FB_KEYS = 4 * FB_WKEYS;             // 384 keys per workgroup
static_assert(FB_KEYS == ATTN_DROP_KWIN, "a fused key block is one row-key window of the dropout mask");
constexpr int FB_QROWS = 64;
constexpr int FB_TILE = FB_QROWS * 128;           // bytes of a 64-row bf16 tile
constexpr int FB_STAGE = 2 * FB_TILE + 2 * FB_QROWS * 4 + (FB_QROWS / 2) * 4;   // Q | dO | -lse*log2e | -delta | dropout row keys (pairs)
constexpr int FB_KIMG = FB_KEYS * 128;
constexpr int FB_SMEM_IMG = 2 * FB_KIMG + 2 * FB_STAGE;
constexpr int FB_SMEM = FB_SMEM_IMG + 16;         // + the readers' counter (FB_SOFT_SYNC)

// ---- dQ across the key blocks of a (sample, head): two forms, chosen per launch (FbWork::handoff).
// ATOMIC (rounds 2-3): every key block adds its [64 x 64] tile to an fp32 [B Lq, H 64] buffer with float atomics (memory-side,
//   ~1.3 TB/s chip-wide, arrival-order dependent in the last bits), a cast kernel rounds the sums to bf16.
// ORDERED HAND-OFF (round 4; cdna_hip_programming.md Appendix B "Attention backward", Guideline 16 recipe R1): the key blocks of a
//   pair form a chain in block order.  Block k reads the running sum of blocks 0..k-1 of a query tile, adds its own tile in
//   registers and stores the new running sum; the LAST block of the pair rounds to bf16 and writes dQ itself.  Plain 16-byte
//   stores and sc1 loads instead of atomics, a fixed summation order - dQ is bit-reproducible - no zero fill and no cast pass.
//   SCOPE OF THE SUMS (round 5): the key blocks of a pair are drawn by workgroups of ONE XCD group (blockIdx % 8, below), i.e. of one
//   XCD and one L2.  The sums therefore never have to leave that L2: the stores are ordinary write-back stores (L1 is write-through,
//   so a store that s_waitcnt vmcnt has retired is in the L2), the loads are sc1 loads (they miss L1 and are served by the L2).
//   With write-through (sc1) stores - the round-4 form - every block's 16 KB per tile went to HBM and came back from it:
//   10.1 GB of HBM traffic per launch at B = 8 against 3.6 GB now, and the launch is 2.8 % faster (profiles/r05_handoff_scope.txt).
//   The premise is CHECKED, not assumed: every workgroup ORs its XCC_ID into a word of its XCD group; a group that sees two
//   different XCDs sets bit 1 of the status word - the step is discarded and the next optimizer call raises, like a timeout
//   (the kernel's WT instantiations / dq_mode bit 9 / T2S_FB_HANDOFF_SCOPE=agent are the write-through form for such a device).  The running sums live in the ACCUMULATOR-NATIVE layout (per tile: wave quadrant, register group, lane: every
//   access a lane-linear 1 KB piece).  Protocol per query tile t (flags[pair][t] = number of blocks whose sum is published):
//     producer  the four waves store their quadrants; one tile later, behind every wave's s_waitcnt vmcnt(0) and the
//               workgroup barrier that phase B needs anyway, ONE lane stores flags[t] = k + 1 (sc1 store)
//     consumer  every wave loads flags[t] (sc1) at the top of tile t, checks it at the end of phase A (spins, bounded, only if
//               the predecessor has not got there yet), then - behind the barrier - loads the sum with sc1 loads
//   A block only ever waits for a block with a SMALLER ticket: workgroups draw their (pair, block) from a per-XCD-group ticket
//   counter (atomic add) in the order they start running, so the block waited for has started, whatever order the hardware
//   dispatches workgroup ids in: no deadlock by construction; the spin is bounded all the same and a timeout is reported in
//   status[0] (the sweep then finishes without waiting: wrong dQ, never a hang).  The tickets number the key blocks that EXIST
//   (FbWork.slots, built by the prep kernel from the key counts), not the [pairs][static bound] rectangle the grid is sized by:
//   the workgroups without a block start last (tools/fused_timeline.py: with them between the chains the CUs were 90 % busy).
struct FbWork {
  float* part;            // hand-off: running sums [B H][nqt][4 quadrants][4 register groups][64 lanes] x 16 B; atomic form: dq32 [B Lq, H 64]
  unsigned* flags;        // [B H][nqt]
  unsigned* tickets;      // [3 launches][8 XCD groups]
  unsigned* status;       // [4]: word 0 bit 0 = a bounded spin timed out, bit 1 = an XCD group ran on more than one XCD (XCD-local sums only)
  const float* nl;        // [B H][nqt * 64]: -lse * log2(e) per query row, -inf behind Lq (written by attn_delta_prep_kernel)
  const float* nd;        // [B H][nqt * 64]: -delta per query row, 0 behind Lq
  const unsigned* slots;  // [8 XCD groups][groups + 1]: first ticket of each (sample, head) pair of the group, then the group's total (prep kernel)
  int groups;             // (sample, head) pairs per XCD group = ceil(B H / 8)
  int handoff;
  unsigned spin_limit;    // polls a hand-off wait may take before it gives up (FB_SPIN_LIMIT; 0 in the diagnostic mode of the tests)
  int never_publish;      // diagnostic mode (dq_mode bit 8): no block publishes its flags - every successor's wait times out
  int diag_misplaced;     // diagnostic mode (dq_mode bit 10): the placement check sees alternating XCDs inside every group
};
// Staging of the Q / dO tiles and their row constants by LDS-DMA (buffer_load ... lds: 1 KB per wave-instruction = 8 rows x 128 B
// straight into the swizzled tile image - the chunk swizzle is a permutation INSIDE a row, so it goes on the per-lane source address;
// rows behind Lq fall outside the descriptor's records and read as zeros - the tile's row offset is part of the VECTOR offset, which the
// hardware range-checks (a scalar offset is not: ADVICE r4); the row constants arrive pre-scaled from the prep kernel).  VERDICT r3 #3;
// cdna_hip_programming.md rule 21.  Dropout masks of the pipelined sweep are KEEP words applied to the fp32 dP values by one
// v_and_b32_sdwa per score, their three packed instructions skewed over the chunks of a block (round 4).  The diagnostic / timing-only
// variants of this kernel (cycle stamps, workgroup timeline, register-staged Q / dO, drop-word masks, the LDS-prefetched hand-off, the
// "results wrong" ablation switches) live in tools/ablate/attn_bwd_fused_bf16_diag.hip, not in the product library.
// The DMA is issued through inline asm: told about an LDS-DMA builtin, the compiler orders every later LDS read whose address it cannot
// prove disjoint (the stage buffer index is a run-time value) behind it with an s_waitcnt vmcnt(0) - in the middle of phase A, a
// full memory round trip per tile (seen in the ISA).  The waits are placed by hand instead: every wave waits for its own pieces
// (s_waitcnt vmcnt(0) in FB_STAGE_WRITE) ahead of the barrier that publishes the buffer; the buffer being filled is not read by
// anyone between the barrier that retired its previous tile and that one.  (M0 = LDS byte address of the piece; one wait state
// between the M0 write and the DMA.)
__device__ __forceinline__ u32x4 fb_rsrc_s(const void* base, uint32_t bytes) {        // descriptor in SCALAR registers, by construction
  const uint64_t a = (uint64_t)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) & 0xffffu;
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ uint32_t fb_lds_addr(const char* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void fb_dma16(u32x4 rs, uint32_t lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void fb_dma16_sc1(u32x4 rs, uint32_t lds, int voff, int soff) {      // handed-off bytes: every load of them bypasses L1 (sc1)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void fb_dma4(u32x4 rs, uint32_t lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
constexpr int FB_CTRL_WORDS = 64;                       // tickets (24) + status (4) + XCDs seen per group (8, at word 32), padded: the block the launch zeroes, with the flags behind it
constexpr int FB_XCC_SEEN = 32;
constexpr unsigned FB_SPIN_LIMIT = 1u << 18;            // ~1-2 us per poll: a few tenths of a second (a legitimate wait is < 1 ms)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t fb_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// ---- MFMAs with the register file of their accumulator chosen by hand.  With 512 registers per lane the compiler selects the
// AGPR form for every MFMA and then copies each S / dP tile between the two files around the softmax (816 v_accvgpr_* in
// the first build of this kernel: more VALU time than the MFMAs themselves).  Here the long-lived dK^T / dV^T accumulators are
// tied to AGPRs ("+a") and the S / dP tiles to VGPRs ("+v"), so no copy exists.  hipcc pads nothing inside or behind an asm
// statement (cdna_hip_programming.md section 5.7): the leading s_nop 1 covers a VALU-written operand, the s_nop 11 behind the
// last MFMA of a VGPR chain covers its result being read by VALU code (8-pass MFMA: 12 wait states).
#define FB_U4(x) __builtin_bit_cast(u32x4, x)
__device__ __forceinline__ void fb_mfma_sdp(f32x16& sacc, f32x16& dpacc, const bf16x8 (&qf)[4], const bf16x8 (&kf)[4], const bf16x8 (&dof)[4],
                                            const u32x4 (&vf)[4]) {
  asm("s_nop 1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %2, %6, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %1, %10, %14, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %3, %7, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %1, %11, %15, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %4, %8, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %1, %12, %16, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %5, %9, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %1, %13, %17, %1\n\t"
      "s_nop 11"
      : "+v"(sacc), "+v"(dpacc)
      : "v"(FB_U4(qf[0])), "v"(FB_U4(qf[1])), "v"(FB_U4(qf[2])), "v"(FB_U4(qf[3])), "v"(FB_U4(kf[0])), "v"(FB_U4(kf[1])), "v"(FB_U4(kf[2])),
        "v"(FB_U4(kf[3])), "v"(FB_U4(dof[0])), "v"(FB_U4(dof[1])), "v"(FB_U4(dof[2])), "v"(FB_U4(dof[3])), "a"(vf[0]), "a"(vf[1]),
        "a"(vf[2]), "a"(vf[3]));        // the V fragments live in AGPRs for the whole kernel (an MFMA B operand may be an AGPR)
}
// dV^T[db] += dO^T[s][db] P[s], dK^T[db] += Q^T[s][db] dS[s]  (s = 0, 1; db = 0, 1): eight MFMAs into AGPR accumulators
__device__ __forceinline__ void fb_mfma_dvdk(f32x16& dv0, f32x16& dv1, f32x16& dk0, f32x16& dk1, const bf16x8 (&doT)[2][2], const bf16x8 (&qT)[2][2],
                                             const bf16x8 (&pf)[2], const bf16x8 (&dsf)[2]) {
  asm("s_nop 1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %4, %12, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %2, %8, %14, %2\n\t"
      "v_mfma_f32_32x32x16_bf16 %1, %5, %12, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %3, %9, %14, %3\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %6, %13, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %2, %10, %15, %2\n\t"
      "v_mfma_f32_32x32x16_bf16 %1, %7, %13, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %3, %11, %15, %3"
      : "+a"(dv0), "+a"(dv1), "+a"(dk0), "+a"(dk1)
      : "v"(FB_U4(doT[0][0])), "v"(FB_U4(doT[0][1])), "v"(FB_U4(doT[1][0])), "v"(FB_U4(doT[1][1])), "v"(FB_U4(qT[0][0])), "v"(FB_U4(qT[0][1])),
        "v"(FB_U4(qT[1][0])), "v"(FB_U4(qT[1][1])), "v"(FB_U4(pf[0])), "v"(FB_U4(pf[1])), "v"(FB_U4(dsf[0])), "v"(FB_U4(dsf[1])));
}

// ---- single-MFMA statements for the software-pipelined sweep (the compiler orders them among the VALU code between the
// scheduling fences; it knows nothing of their latency, so every consumer is placed by construction - see FB_PIPE below)
#define FB_MFMA_V0(acc, a, b, c) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(FB_U4(a)), "v"(FB_U4(b)), "v"(c))
#define FB_MFMA_V(acc, a, b) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(FB_U4(a)), "v"(FB_U4(b)))
#define FB_MFMA_VA0(acc, a, ba, c) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(FB_U4(a)), "a"(ba), "v"(c))
#define FB_MFMA_VA(acc, a, ba) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(FB_U4(a)), "a"(ba))
#define FB_MFMA_VAZ(acc, a, ba) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(FB_U4(a)), "a"(ba))
// (no s_nop: the P / dS operand words are written by VALU code at least one MFMA group ahead of the MFMA that reads them)
#define FB_MFMA_A(acc, a, b) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(FB_U4(a)), "v"(b))
#define FB_FENCE() __builtin_amdgcn_sched_barrier(0)

// transposed fragment from a row-block base and this lane's two precomputed offsets (rows r and r + 8 of the block)
__device__ __forceinline__ bf16x8 fb_tr(const char* base, const int (&va2)[2]) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + va2[0]));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + va2[1]));
  const s16x8 c = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, c);
}

// the same from two ABSOLUTE LDS byte addresses held in vector registers plus a compile-time offset: the offset rides in the instruction's
// 16-bit offset field (base + constant in a scalar register, as the compiler forms it for fb_tr, costs a v_add_u32 per read)
__device__ __forceinline__ bf16x8 fb_tr_abs(const int (&abs2)[2], const int off) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(uint32_t)(abs2[0] + off));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(uint32_t)(abs2[1] + off));
  const s16x8 c = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, c);
}

__device__ __forceinline__ uint32_t fb_pack2(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, bf16x2_t{(__bf16)a, (__bf16)b});
}

// m-th MFMA (m = 0..7) of G1(block I): S and dP chains of key block I % 3, alternating; each chain starts from the row constants
template <int I, int M, bool DROP>
__device__ __forceinline__ void fb_g1(f32x16 (&sacc)[2], f32x16 (&dpacc)[2], const bf16x8 (&qf)[4], const bf16x8 (&dof)[4], const bf16x8 (&kf)[4],
                                      const u32x4 (&vf)[FB_KB][4]) {
  constexpr int s = M / 2, kb = I % 3, par = I & 1;
  if constexpr (M % 2 == 0) FB_MFMA_V(sacc[par], qf[s], kf[s]);
  else if constexpr (DROP && s == 0) FB_MFMA_VAZ(dpacc[par], dof[0], vf[kb][0]);      // dropout: dP from zero, delta subtracted behind the mask
  else FB_MFMA_VA(dpacc[par], dof[s], vf[kb][s]);
}
// m-th MFMA of G2(block I): m = 0..3 dV^T (needs P), m = 4..7 dK^T (needs dS)
template <int I, int M>
__device__ __forceinline__ void fb_g2(f32x16 (&dvacc)[FB_KB][2], f32x16 (&dkacc)[FB_KB][2], const bf16x8 (&doT)[2][2], const bf16x8 (&qT)[2][2],
                                      const uint32_t (&pfw)[8], const uint32_t (&dsw)[8]) {
  constexpr int kb = I % 3, s = (M & 3) >> 1, db = M & 1;
  if constexpr (M < 4) {
    const u32x4 b = {pfw[4 * s], pfw[4 * s + 1], pfw[4 * s + 2], pfw[4 * s + 3]};
    FB_MFMA_A(dvacc[kb][db], doT[s][db], b);
  } else {
    const u32x4 b = {dsw[4 * s], dsw[4 * s + 1], dsw[4 * s + 2], dsw[4 * s + 3]};
    FB_MFMA_A(dkacc[kb][db], qT[s][db], b);
  }
}
// chunk m (registers 2m, 2m+1) of the softmax of block I: P = exp2(S'), packed bf16 operand word.  EDGE: the validity /
// decoder rule - this lane's key is visible to the tile rows >= thr (thr = first visible row - row of register 0 of this lane;
// register r is row (r & 3) + 8 (r >> 2) above it), so one compare + select per score
template <int I, int M, bool EDGE, bool DROP>
__device__ __forceinline__ void fb_ve(f32x16 (&sacc)[2], uint32_t (&pfw)[8], const int thr, uint32_t (&mw)[8], const uint32_t (&rkw)[8], const uint32_t ck2,
                                      const uint32_t th2) {
  constexpr int par = I & 1, r0 = 2 * M, r1 = 2 * M + 1;
  float p0 = fast_exp2(sacc[par][r0]), p1 = fast_exp2(sacc[par][r1]);
  if (EDGE) {
    p0 = ((r0 & 3) + 8 * (r0 >> 2)) >= thr ? p0 : 0.f;
    p1 = ((r1 & 3) + 8 * (r1 >> 2)) >= thr ? p1 : 0.f;
  }
  sacc[par][r0] = p0;                                  // dS uses the UNdropped probability
  sacc[par][r1] = p1;
  if (DROP) {                                          // registers (2m, 2m+1) are two consecutive queries of this lane's key: one packed mask word
    // keep word of chunk M: 0xFFFF in every KEPT half (th2 = attn_drop_thresh2k).  Its three packed instructions (multiply, saturating
    // subtract, shift) are SKEWED over the chunks of the block - chunk M + 2 is multiplied, M + 1 subtracted and M shifted in one group -
    // so that no packed instruction stands right behind the one it depends on (each such pair costs an s_nop in an issue-bound gap)
    if constexpr (M == 0) {
      mw[0] = attn_drop_kept_mul(rkw[0], ck2); mw[1] = attn_drop_kept_mul(rkw[1], ck2); mw[2] = attn_drop_kept_mul(rkw[2], ck2);
      mw[0] = attn_drop_kept_sub(mw[0], th2); mw[1] = attn_drop_kept_sub(mw[1], th2);
    } else {
      if constexpr (M + 2 < 8) mw[M + 2] = attn_drop_kept_mul(rkw[M + 2], ck2);
      if constexpr (M + 1 < 8) mw[M + 1] = attn_drop_kept_sub(mw[M + 1], th2);
    }
    mw[M] = attn_drop_kept_mask(mw[M]);
    pfw[M] = fb_pack2(p0, p1) & mw[M];                 // dV uses the dropped one (scaled by 1/(1-p) at the end)
  } else {
    pfw[M] = fb_pack2(p0, p1);
  }
}
// chunks m, m + 1 of dS of block I: P * dP' (dP' = dP - delta from the seeded chain), or with dropout P * (dP * M / (1-p) - delta).
// The two scores of a chunk sit in an even-aligned register pair (accumulator registers 2m, 2m + 1), and so do their row constants:
// the fp32 arithmetic runs as PACKED instructions (v_pk_fma_f32 / v_pk_mul_f32: two fp32 lanes per issue slot, the same IEEE results
// as the scalar forms) - one VALU slot per score instead of two (round 6: one wave per SIMD issues in order, every slot counts).
// A packed instruction right behind the one it depends on costs a wait state (the compiler pads an s_nop), so the chunks go through
// in PAIRS, stage by stage - mask, mask, fma, fma, mul, mul, cvt, cvt - and the pair may be cut in two (HEAD: up to the fma; TAIL:
// the rest) so that a slot's single-chunk MFMA groups carry half a pair each.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int I, int M, bool DROP, bool HEAD, bool TAIL_>
__device__ __forceinline__ void fb_vm2(const f32x16 (&sacc)[2], const f32x16 (&dpacc)[2], uint32_t (&dsw)[8], const uint32_t (&mw)[8], const f32x4 (&dl)[4],
                                       const float inv, f32x2 (&t)[2]) {
  constexpr int par = I & 1;
  static_assert((M & 1) == 0, "chunk pairs");
  if constexpr (HEAD) {
    if (DROP) {
      // (one v_and_b32_sdwa per score: the kept-half word, sign-extended by the operand selector, is the fp32 mask)
      const f32x2 d0 = {attn_drop_keep_lo(dpacc[par][2 * M], mw[M]), attn_drop_keep_hi(dpacc[par][2 * M + 1], mw[M])};
      const f32x2 d1 = {attn_drop_keep_lo(dpacc[par][2 * M + 2], mw[M + 1]), attn_drop_keep_hi(dpacc[par][2 * M + 3], mw[M + 1])};
      t[0] = __builtin_elementwise_fma(d0, f32x2{inv, inv}, f32x2{dl[M >> 1][0], dl[M >> 1][1]});
      t[1] = __builtin_elementwise_fma(d1, f32x2{inv, inv}, f32x2{dl[M >> 1][2], dl[M >> 1][3]});
    } else {
      t[0] = f32x2{dpacc[par][2 * M], dpacc[par][2 * M + 1]};
      t[1] = f32x2{dpacc[par][2 * M + 2], dpacc[par][2 * M + 3]};
    }
  }
  if constexpr (TAIL_) {
    const f32x2 r0 = f32x2{sacc[par][2 * M], sacc[par][2 * M + 1]} * t[0];
    const f32x2 r1 = f32x2{sacc[par][2 * M + 2], sacc[par][2 * M + 3]} * t[1];
    dsw[M] = fb_pack2(r0[0], r0[1]);
    dsw[M + 1] = fb_pack2(r1[0], r1[1]);
  }
}

// MODE 3 (round 4, the shipped launch): BOTH kinds of workgroup in ONE launch - one workgroup-uniform branch at the top picks the
// straight-line sweep or the edge sweep.  As two launches the edge blocks (one per (sample, head): 768 workgroups at B = 64, each a
// full query sweep for half a block of keys on average) ran alone on the chip for 1.6 - 2.3 ms per call (3 waves of workgroups on 256
// CUs, 7 - 10 % of the fused backward); in one launch they fill in between the other workgroups.  MODE 0 / 1 remain for A/B runs
// (T2S_FB_SPLIT_EDGE=1).
// MODE 0: workgroups whose 384 keys are all valid prefix keys run the software-pipelined sweep, the others exit; MODE 1: the
// complement (the same pipeline with the validity / decoder rule applied to P); MODE 2:
// the tail launch (plain sweep, loops over the key blocks beyond the static bound).  Separate kernels, so that each is
// register-allocated for one sweep.
// WT: the hand-off's running sums are stored write-through (sc1) and the placement of the XCD groups is not looked at (dq_mode bit 9);
// a compile-time choice - a run-time one would be a branch inside a slot of the interleaved sweep (see ILV below), and issuing both
// stores through two descriptors, one of them empty, measured ~1 % slower than one store.  Instantiated for the shipped launch
// forms only (MODE 3 and the tail launch).
template <bool USE_IDX, int MODE, bool DROP, bool HO, bool WT = false>
__global__ __launch_bounds__(256, 1) void attn_bwd_fused_bf16_kernel(AttnParams p, FbWork w) {
  constexpr bool TAIL = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const stage = smem;                        // 2 x (Q tile | dO tile | -lse*log2e | -delta)
  char* const kimg = smem + 2 * FB_STAGE;          // [384 keys][64 d] bf16, tile_off swizzle
  char* const dsimg = kimg + FB_KIMG;              // [384 keys][64 q] bf16, same layout
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  float* __restrict__ const dq32 = w.part;         // (atomic form)
  int kblk, h, b;
  if constexpr (HO && !TAIL) {
    // ticket: this workgroup is the slot-th one of its XCD group to START (not the slot-th by id), see FbWork
    // Tickets are COMPACT: slot s of XCD group x is the s-th key block that exists in the group's pairs (table of first slots per
    // pair, built by the prep kernel from the key counts), not the s-th of a [pairs][kblocks] rectangle sized by the static key
    // bound.  The workgroups the grid has beyond a group's total leave at once - and they are the LAST to start, where the
    // rectangle had up to kblocks - 1 of them in a row between two pairs, each holding a whole CU for a launch, an atomic round
    // trip and a count load (tools/fused_timeline.py: CUs 90 % busy over the launch, median gap between two sweeps 12 us).
    unsigned* sl = reinterpret_cast<unsigned*>(smem);
    const int xg = (int)(blockIdx.x % T2S_XCDS);
    const unsigned* __restrict__ tab = w.slots + xg * (w.groups + 1);
    if (tid == 0) {
      unsigned* tk = w.tickets + (MODE % 3) * T2S_XCDS + xg;
      // (a stale read only errs on the low side: then the atomic decides)
      sl[0] = __hip_atomic_load(tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= tab[w.groups] ? 0xFFFFFFFFu : atomicAdd(tk, 1u);
      if constexpr (!WT) {
        // XCD-local sums: every workgroup of this group must sit on the same XCD.  The first one to find another XCD's bit reports it.
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        if (w.diag_misplaced) xcc = (blockIdx.x / T2S_XCDS) & 1u;
        const unsigned seen = atomicOr(w.tickets + FB_XCC_SEEN + xg, 1u << xcc);
        if (seen & ~(1u << xcc)) atomicOr(w.status, 2u);
      }
    }
    __syncthreads();
    const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane((int)sl[0]);
    __syncthreads();                               // (the stage buffer is written below)
    if (slot >= tab[w.groups]) return;
    int g = 0;                                     // the pair whose slot range holds this ticket: count the first-slots <= slot
    for (int base = 0; base < w.groups; base += 64) {
      const int gi = base + lane;
      const unsigned first = gi < w.groups ? tab[gi] : 0xFFFFFFFFu;
      g += __builtin_popcountll(__ballot(first <= slot));
    }
    g = __builtin_amdgcn_readfirstlane(g) - 1;
    const int bh = g * T2S_XCDS + xg;
    // (the division runs on the vector ALU: pin the results to scalar registers, or every address and the buffer descriptors
    // derived from them live in VGPRs and each buffer access becomes a waterfall loop - cdna_hip_programming.md T20)
    kblk = __builtin_amdgcn_readfirstlane((int)(slot - tab[g]));
    b = __builtin_amdgcn_readfirstlane(bh / p.H);
    h = __builtin_amdgcn_readfirstlane(bh - b * p.H);
  } else {
    if (!attn_xcd_tile(TAIL ? 1 : p.kblocks, p.H, p.B, kblk, h, b)) return;       // workgroup-uniform (attn_common.h)
  }
  const int n_prefix = USE_IDX ? p.kv_cnt[b] : (p.idx_cap - p.n_dec);
  const int nk = n_prefix + p.n_dec;
  int kbw = TAIL ? p.kblocks : kblk;
  if (kbw * FB_KEYS >= nk) return;     // uniform per workgroup
  const int32_t* __restrict__ idx = USE_IDX ? p.kv_idx + (int64_t)b * p.idx_cap : nullptr;
  const bf16_t* __restrict__ Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.q_bs + h * 64;
  const bf16_t* __restrict__ DO = reinterpret_cast<const bf16_t*>(p.dout) + (int64_t)b * p.o_bs + h * 64;
  const char* __restrict__ Kg = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.kv_bs + h * 64);
  const char* __restrict__ Vg = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.kv_bs + h * 64);
  const float* __restrict__ LSE = p.lse + ((int64_t)b * p.H + h) * p.Lq;
  const float* __restrict__ DELTA = p.delta + ((int64_t)b * p.H + h) * p.Lq;
  float* __restrict__ DQ = dq32 + (int64_t)b * p.Lq * (p.H * 64) + h * 64;
  const float c = p.scale * LOG2E;
  const int nqt = (p.Lq + FB_QROWS - 1) / FB_QROWS;
  const int64_t pair = (int64_t)b * p.H + h;
  unsigned* const flags_pair = HO ? w.flags + pair * nqt : nullptr;
  const char* const part_pair = HO ? reinterpret_cast<const char*>(w.part) + pair * nqt * (int64_t)(FB_QROWS * 64 * 4) : nullptr;
  const int sr = tid >> 3, sc = tid & 7;
  // per-lane LDS byte offsets, computed once: row fragment of row lr of a 32-row block (chunk 2s + lh), transposed fragment
  // (rows 4lh + qq and + 8, chunk 4db + 2g1 + (pp >> 1)), dS^T store (row lr, chunk cc); block / tile / image offsets are
  // immediates or uniform adds on top of these
  int ka[4], va[2][2], vads[2][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) ka[s] = tile_off(lr, 2 * s + lh);
  {
    const int g1 = (lane >> 4) & 1, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int chunk = 4 * db + 2 * g1 + (pp >> 1);
      va[db][0] = tile_off(4 * lh + qq, chunk) + ((pp & 1) << 3);
      va[db][1] = tile_off(4 * lh + qq + 8, chunk) + ((pp & 1) << 3);
      // the dS^T image swaps the two 8-byte halves of every 16-byte chunk on ODD rows (see the stores below): the same involution here
      vads[db][0] = tile_off(4 * lh + qq, chunk) + (((pp ^ qq) & 1) << 3);
      vads[db][1] = tile_off(4 * lh + qq + 8, chunk) + (((pp ^ qq) & 1) << 3);
    }
  }
  const int wrow = lr * 128, wxor = tile_f(lr) << 4;          // dS^T store of chunk cc: wrow + ((cc << 4) ^ wxor)

  do {   // key blocks of this workgroup (TAIL == false: exactly one, no loop is compiled)
    const int kp0 = kbw * FB_KEYS;
    const int nkeys_wg = (nk - kp0) < FB_KEYS ? (nk - kp0) : FB_KEYS;              // valid keys of this workgroup
    const int nks = (nkeys_wg + 15) >> 4;                                          // 16-key steps of the dQ product
    // decoder keys or the end of the list inside; hand-off: the LAST block of a pair (it writes bf16 dQ) is always an edge block,
    // also when the list ends exactly on its boundary, so that the straight-line MODE 0 sweep has one output form only
    const bool edge_wg = (kp0 + FB_KEYS > n_prefix) || (HO && kp0 + FB_KEYS >= nk);
    if (MODE == 0 && edge_wg) return;                                              // (nkeys_wg == FB_KEYS follows from !edge_wg)
    if (MODE == 1 && !edge_wg) return;
    // hand-off chain position (workgroup-uniform): block 0 has no predecessor, the block that holds the end of the list finishes dQ
    const bool ho_last = HO && (MODE != 0) && (kp0 + FB_KEYS >= nk);              // (a last block is an edge block: never in the FULL, non-EDGE sweep)
    int ho_wait = (HO && !TAIL) ? kbw : 0;                                         // flags[t] must reach this before tile t's sum is read
    float ho_ln2 = 0.6931471805599453f;                                            // dQ = ln 2 * acc (K is pre-scaled by log2 e); NaN once a wait has timed out
    // running sums of this pair through a buffer descriptor: block 0 gets ZERO records - its loads return 0.0 without touching
    // memory, so the sweep needs no branch around them
    const __amdgpu_buffer_rsrc_t rs_ld = fb_rsrc(part_pair, (HO && kbw > 0) ? (unsigned)(nqt * (FB_QROWS * 64 * 4)) : 0u);
    const __amdgpu_buffer_rsrc_t rs_st = fb_rsrc(part_pair, HO ? (unsigned)(nqt * (FB_QROWS * 64 * 4)) : 0u);
    // ---- K image of the workgroup's keys, pre-scaled by scale*log2e (one bf16 rounding per element, as the dK/dV kernel's
    // register fragments); rows past the list repeat its last key (their P is forced to 0 below)
#pragma unroll
    for (int i = 0; i < FB_KEYS / 32; ++i) {
      const int row = sr + 32 * i;
      int kp = kp0 + row;
      kp = kp < nk ? kp : nk - 1;
      const uint32_t grow = USE_IDX ? (uint32_t)idx[kp] : (uint32_t)kp;
      bf16x8 kv = *reinterpret_cast<const bf16x8*>(Kg + ((size_t)grow * (size_t)p.kv_rs + (size_t)sc * 8) * 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) kv[j] = (bf16_t)((float)kv[j] * c);
      *reinterpret_cast<bf16x8*>(kimg + tile_off(row, sc)) = kv;
    }
    // ---- this wave's keys: V fragments (B operands of dP), list positions, validity
    u32x4 vf[FB_KB][4];
    int kdec[FB_KB];          // decoder step of this lane's key of block kb (negative: prefix key)
    bool kvalid[FB_KB];
    int64_t krow[FB_KB];
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb) {
      const int kpos = kp0 + wave * FB_WKEYS + kb * 32 + lr;
      kvalid[kb] = kpos < nk;
      const int kc = kvalid[kb] ? kpos : nk - 1;
      krow[kb] = USE_IDX ? (int64_t)idx[kc] : (int64_t)kc;
      kdec[kb] = kpos - n_prefix;
      const char* vp = Vg + (krow[kb] * p.kv_rs + 8 * lh) * 2;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        vf[kb][s] = *reinterpret_cast<const u32x4*>(vp + 32 * s);
        asm volatile("" : "+a"(vf[kb][s]));          // park it in the accumulator file
      }
    }
    f32x16 dkacc[FB_KB][2], dvacc[FB_KB][2];
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dkacc[kb][0][i] = 0.f; dkacc[kb][1][i] = 0.f; dvacc[kb][0][i] = 0.f; dvacc[kb][1][i] = 0.f; }

    // attention-probability dropout (attn_common.h): this lane's column key of each key block in both 16-bit halves; the row keys of
    // the tile's 32 query pairs are hashed by threads 0..31 while the tile is staged (dkdv kernel's scheme: the same mask function)
    const uint32_t salt = DROP ? attn_drop_salt(p.drop_seed_lo, p.drop_seed_hi, (uint32_t)(b * p.H + h)) : 0u;
    uint32_t ck2[FB_KB];
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb) ck2[kb] = 0u;         // (set per 256-row window of query rows at the top of every fourth tile)
    const uint32_t th2 = attn_drop_thresh2k(p.drop_thresh);      // (the pipelined sweep's masks are KEEP words)
    const float drop_inv = p.drop_inv;
    uint32_t rkreg = 0;
    int ld_row0 = 0;              // first query row of the tile being loaded (uniform)
    // ---- staging of the Q / dO tiles by LDS-DMA.  A tile is 8 pieces of 8 rows; wave w issues pieces w and w + 4 of Q and of dO.
    // Lane: row 8 piece + lane / 8, chunk POSITION lane % 8, which holds the logical chunk (lane % 8) ^ tile_f(row); tile_f depends on
    // bits 1..3 of the row, i.e. on lane / 8 and on the parity of the piece - the same for pieces w and w + 4: ONE offset per operand
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int st_r8 = lane >> 3;
    const int st_chunk = (lane & 7) ^ tile_f(wave_s * 8 + st_r8);
    const int voff_q = st_r8 * (int)p.q_rs * 2 + (st_chunk << 4), voff_o = st_r8 * (int)p.o_rs * 2 + (st_chunk << 4);
    const u32x4 rs_q = fb_rsrc_s(Q, (unsigned)(((int64_t)(p.Lq - 1) * p.q_rs + 64) * 2));
    const u32x4 rs_o = fb_rsrc_s(DO, (unsigned)(((int64_t)(p.Lq - 1) * p.o_rs + 64) * 2));
    const u32x4 rs_nl = fb_rsrc_s(w.nl + pair * nqt * FB_QROWS, (unsigned)(nqt * FB_QROWS * 4));
    const u32x4 rs_nd = fb_rsrc_s(w.nd + pair * nqt * FB_QROWS, (unsigned)(nqt * FB_QROWS * 4));
    const uint32_t st_lds = __builtin_amdgcn_readfirstlane(fb_lds_addr(stage));
    const int q_rs2 = __builtin_amdgcn_readfirstlane((int)p.q_rs * 2), o_rs2 = __builtin_amdgcn_readfirstlane((int)p.o_rs * 2);
    int st_buf = 0;               // stage buffer the next FB_STAGE_LOAD fills (uniform)
#define FB_STAGE_LOAD()                                                                         \
  {                                                                                             \
    const uint32_t dst_ = st_lds + (uint32_t)(st_buf * FB_STAGE + wave_s * 1024);               \
    const int r0_ = ld_row0 + wave_s * 8;                                                       \
    /* the tile's row offset rides in the VECTOR offset: that one is range-checked against the descriptor (rows behind Lq read 0) */ \
    fb_dma16(rs_q, dst_, voff_q + r0_ * q_rs2, 0);                                              \
    fb_dma16(rs_o, dst_ + FB_TILE, voff_o + r0_ * o_rs2, 0);                                    \
    fb_dma16(rs_q, dst_ + 4096, voff_q + (r0_ + 32) * q_rs2, 0);                                \
    fb_dma16(rs_o, dst_ + FB_TILE + 4096, voff_o + (r0_ + 32) * o_rs2, 0);                      \
    if (wave_s == 0) fb_dma4(rs_nl, st_lds + (uint32_t)(st_buf * FB_STAGE + 2 * FB_TILE), lane * 4, ld_row0 * 4);                   \
    if (wave_s == 1) fb_dma4(rs_nd, st_lds + (uint32_t)(st_buf * FB_STAGE + 2 * FB_TILE + FB_QROWS * 4), lane * 4, ld_row0 * 4);    \
    if (DROP && tid < FB_QROWS / 2) {                                                           \
      const int qa_ = ld_row0 + 2 * tid, qb2_ = qa_ + 1;                                        \
      rkreg = attn_drop_rowkey16(salt, qa_ < p.Lq ? qa_ : p.Lq - 1, kbw) | (attn_drop_rowkey16(salt, qb2_ < p.Lq ? qb2_ : p.Lq - 1, kbw) << 16); /* window = this key block */ \
    }                                                                                           \
    ld_row0 += FB_QROWS;                                                                        \
  }
    // the DMA pieces of the tile must have landed before the barrier that publishes the buffer: every wave waits for its own
#define FB_STAGE_WRITE(buf_)                                                                    \
  {                                                                                             \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
    if (DROP && tid < FB_QROWS / 2) reinterpret_cast<uint32_t*>(stage + (buf_) * FB_STAGE + 2 * FB_TILE + 2 * FB_QROWS * 4)[tid] = rkreg; \
    st_buf ^= 1;                                                                                \
  }
#define FB_FLAG_WAIT(addr_, fv_)      /* bounded spin until *addr_ >= ho_wait (fv_: a value already read from it) */  \
  {                                                                                             \
    int fvs_ = __builtin_amdgcn_readfirstlane((int)(fv_));                                      \
    if (fvs_ < ho_wait) {                                                   \
      unsigned spins_ = 0;                                                                      \
      do {                                                                                      \
        __builtin_amdgcn_s_sleep(16);                                                           \
        fvs_ = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load((addr_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); \
        if (++spins_ > w.spin_limit) {      /* never in a correct run: report, stop waiting and POISON what this block hands on - */ \
          if (lane == 0) atomicOr(w.status, 1u);      /* its sums, hence the pair's dQ rows, become NaN: no hang, no silent error */   \
          ho_wait = 0;                                                                          \
          ho_ln2 = __builtin_nanf("");                                                          \
        }                                                                                       \
      } while (fvs_ < ho_wait);                                                                 \
    }                                                                                           \
  }
    unsigned* const rd_cnt = reinterpret_cast<unsigned*>(smem + FB_SMEM_IMG);       // FB_SOFT_SYNC: waves through slot 1, over all tiles so far
    if (tid == 0) *rd_cnt = 0u;
    FB_STAGE_LOAD();
    FB_STAGE_WRITE(0);
    __syncthreads();

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int dq_qb = wave_u >> 1, dq_db = wave_u & 1;          // this wave's 32 x 32 tile of dQ (provably wave-uniform)
    // this wave's two operand address pairs of the dQ product, as ABSOLUTE LDS byte addresses (dS^T image / K image + the lane's offsets):
    // a step's constant then rides in the read's 16-bit offset field (scalar base + constant and a vector offset: one v_add_u32 per read).
    // FB_ABS_HERE() re-defines them, as far as the compiler can tell, in the basic block that reads through them: "address + constant" is
    // loop-invariant, and hoisted out of the sweep (or merely out of the block: instruction selection folds offsets per block) the 84
    // sums of a tile's dQ reads take registers this kernel does not have - they came back as one v_add_u32 per read, or from scratch.
#define FB_ABS_HERE() asm("" : "+v"(vaq_abs[0]), "+v"(vaq_abs[1]), "+v"(vad_abs[0]), "+v"(vad_abs[1]))
    int vaq_abs[2] = {(int)fb_lds_addr(dsimg) + (dq_qb ? vads[1][0] : vads[0][0]), (int)fb_lds_addr(dsimg) + (dq_qb ? vads[1][1] : vads[0][1])};
    int vad_abs[2] = {(int)fb_lds_addr(kimg) + (dq_db ? va[1][0] : va[0][0]), (int)fb_lds_addr(kimg) + (dq_db ? va[1][1] : va[0][1])};
    // EDGE: first tile row that sees this lane's key of block kb (prefix keys: every row; keys past the list: none)
    int qmin[FB_KB];
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb) qmin[kb] = !kvalid[kb] ? (1 << 30) : (kdec[kb] < 0 ? -(1 << 30) : p.dec_q0 + kdec[kb]);
    // the query sweep in two compiled forms behind ONE workgroup-uniform branch: FULL = all 384 keys valid prefix keys (no
    // validity / decoder rule, no skipped key blocks: straight-line code), and the general form
    auto sweep = [&](auto pipe_tag, auto edge_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(pipe_tag)::value;         // the software-pipelined phase A (all three key blocks of every wave run)
    constexpr bool EDGE = decltype(edge_tag)::value;         // ... with the validity / decoder rule applied to P
    constexpr bool PREF = FULL && !EDGE;                     // the first operands of tile t+1 are fetched from LDS behind the barrier that ends tile t
    // ILV (round 5): the dQ product ("phase B") of tile t - 1 runs INSIDE slots 0 and 1 of phase A of tile t, twelve of its 24 MFMAs
    // in each, every one followed by the transposed reads of the step four ahead.  The dS^T image stays single: the barrier at the end of
    // phase A of tile t - 1 completes it, the bare s_barrier behind slot 1 of tile t ("every wave is done reading the dS^T image") - which
    // the serial form has too - retires its readers ahead of the first dS^T store of tile t in slot 2.  What goes away is the serial
    // phase B (2 024 cycles of LDS-bound MFMAs with the vector ALU idle) and nothing is added to the 6-block pipeline's ramp: the
    // 256-key variant (tools/ablate/attn_bwd_fused_bf16_ilv256.hip) lost exactly there.  The edge sweep and the tail launch keep the serial form.
    // Measured, same box, interleaved runs.  With WRITE-THROUGH running sums the interleave bought nothing: 11 % fewer cycles per tile,
    // the clock down from 2.28 to 2.03 GHz (profiles/r05_fused_ilv384_vs_serial_ab.txt, r05_fused_ilv384_stamps.txt) - the chip gave the
    // cycles back.  With the sums in the XCD's L2 (1.9 TB/s of fabric traffic gone) it is 2.6 - 3.5 % faster than the serial form at
    // B = 32 (21.4 vs 22.0 ms; 18.8 vs 19.2 without dropout) and the B = 64 step 3.7 ms shorter (profiles/r05_fused_ilv384_l2.txt); the
    // serial form is kept as tools/ablate/variants/attn_bwd_fused_bf16_serial384.hip.  Two rules the form depends on, both met the hard way:
    // NO BRANCH inside slots 0 / 1 (the compiler sinks the slot's exponentials out of the MFMA shadow behind it: the hand-off of "tile -1"
    // is done with out-of-range buffer offsets instead, and the scope of the sums' stores is the template parameter WT), and twelve wait
    // states behind the last dQ MFMA on EVERY path (its destination registers are free for the compiler the moment the asm statement has
    // issued; on tile 0, where the result is dropped, the next VALU results landed in them and were overwritten by the MFMA's late write).
    constexpr bool ILV = PREF;
    constexpr bool CAN_LAST = EDGE || !FULL;                 // hand-off: only an edge block (or the tail launch) can end a pair's chain
    bf16x8 qf[4], dof[4], kf[4];
    f32x16 sacc[2], dpacc[2];
    // operands of the dQ product in flight, FB_DQ_DEPTH 16-key steps ahead.  3: measured against 5, 6 and 8 in round 6 (-DFB_DQ_DEPTH=n, same box,
    // interleaved: 20.43 | 20.40 | 20.43 | 21.03 ms at B = 32; 8 spills) - the reads' cost in slots 0 / 1 is LDS occupancy, not their latency
    bf16x8 afA[FB_DQ_DEPTH], bfA[FB_DQ_DEPTH];
    u32x4 pin[4];                                  // hand-off: the predecessor's running sum of the tile whose dQ is being formed
    f32x16 dqacc;
    const char* kw_ = kimg + wave * (FB_WKEYS * 128);
#define FB_LD_QF(qbase_, dobase_, sb_)                                                              \
  _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                   \
    qf[s] = *reinterpret_cast<const bf16x8*>((qbase_) + ka[s] + (sb_) * 4096);                      \
    dof[s] = *reinterpret_cast<const bf16x8*>((dobase_) + ka[s] + (sb_) * 4096);                    \
  }
#define FB_LD_SEEDS(lse_, del_, i_)  /* accumulators of block i start from the row constants of this lane's rows (broadcast reads) */  \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                   \
    const f32x4 l4 = *reinterpret_cast<const f32x4*>((lse_) + ((i_) / 3) * 32 + 8 * g + 4 * lh);    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) sacc[(i_) & 1][4 * g + j] = l4[j];                \
    if (!DROP) {      /* dropout: the dP chain starts from zero, delta is subtracted behind the mask (FB_M) */  \
      const f32x4 d4 = *reinterpret_cast<const f32x4*>((del_) + ((i_) / 3) * 32 + 8 * g + 4 * lh); \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) dpacc[(i_) & 1][4 * g + j] = d4[j];             \
    }                                                                                               \
  }
#define FB_LD_KF(kb_)                                                                               \
  _Pragma("unroll") for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const bf16x8*>(kw_ + ka[s] + (kb_) * 4096);
    if (PREF) {          // tile 0's first operands (stage buffer 0 was written and fenced by the barrier above)
      FB_LD_QF(stage, stage + FB_TILE, 0);
      FB_LD_SEEDS(reinterpret_cast<const float*>(stage + 2 * FB_TILE), reinterpret_cast<const float*>(stage + 2 * FB_TILE) + FB_QROWS, 0);
      FB_LD_KF(0);
    }
    if constexpr (ILV) {   // (slots 0 / 1 of tile 0 multiply operands of a tile that does not exist: defined values, result dropped)
#pragma unroll
      for (int u = 0; u < FB_DQ_DEPTH; ++u) { afA[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; bfA[u] = afA[u]; }
#pragma unroll
      for (int g = 0; g < 4; ++g) pin[g] = u32x4{0u, 0u, 0u, 0u};
    }
    // dQ of query tile tq leaves this block: dqacc (= c dS K over the block's keys; dQ = acc * ln 2) joins the running sum of the pair's
    // earlier blocks (hand-off: stored write-through for the successor, or - last block of the pair - rounded to bf16 and written as
    // dQ), or goes to the fp32 buffer by atomics.  Register r = query row acc_row(r, lh), 32 consecutive dims per half wave: two
    // 128-byte segments per wave instruction.
#define FB_DQ_FINALIZE(tq_)                                                                         \
  {                                                                                                 \
    asm volatile("s_nop 11" : "+v"(dqacc));              /* MFMA result -> VALU read */              \
    const int q0 = (tq_) * FB_QROWS + dq_qb * 32;                                                   \
    if constexpr (HO) {                                                                             \
      const unsigned ho_off = (unsigned)(((tq_) * 4 + wave_u) * 4096 + lane * 16);                  \
      f32x16 tot;                                                                                   \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                              \
        /* (bit_cast of the WHOLE vector, then the element: bit_cast(float, pin[g][j]) is narrowed by this clang to a one-dword load */ \
        /* whose value stands in for all four elements - seen in the ISA) */                        \
        const f32x4 pf = __builtin_bit_cast(f32x4, pin[r >> 2]);                                    \
        tot[r] = __builtin_fmaf(dqacc[r], ho_ln2, pf[r & 3]);                                       \
      }                                                                                             \
      if (CAN_LAST && ho_last) {                                                                    \
        /* the last block of the pair: dQ rows in bf16.  Lanes 2i / 2i+1 hold columns 2i / 2i+1 of the same rows: the even lane */   \
        /* takes the odd lane's value of register 2m, the odd lane the even lane's value of register 2m+1 (DPP quad_perm 1,0,3,2), */ \
        /* each then stores ONE 4-byte pair of its own row */                                       \
        bf16_t* dqp = reinterpret_cast<bf16_t*>(p.dq) + (int64_t)b * p.q_bs + h * 64 + dq_db * 32 + (lr & ~1);                       \
        const bool odd = lr & 1;                                                                    \
        _Pragma("unroll") for (int m = 0; m < 8; ++m) {                                             \
          const float a = tot[2 * m], bb = tot[2 * m + 1];                                          \
          const float send = odd ? a : bb;                                                          \
          const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true)); \
          const int row = q0 + acc_row(2 * m, lh) + (odd ? 1 : 0);                                  \
          if (row < p.Lq) *reinterpret_cast<uint32_t*>(dqp + (int64_t)row * p.q_rs) = odd ? fb_pack2(recv, bb) : fb_pack2(a, recv);   \
        }                                                                                           \
      } else {                                                                                      \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                             \
          const f32x4 t4 = {tot[4 * g], tot[4 * g + 1], tot[4 * g + 2], tot[4 * g + 3]};            \
          /* plain store: the line stays in this XCD's L2;  WT: sc1, write-through */                \
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t4), rs_st, ho_off + g * 1024, 0, WT ? 16 : 0);           \
        }                                                                                           \
      }                                                                                             \
    } else {                                                                                        \
      const int rstep = p.H * 64;                                                                   \
      float* rowp = DQ + (int64_t)q0 * rstep + dq_db * 32;          /* wave-uniform (dq_qb / dq_db come from readfirstlane) */       \
      const int loff = lr + 4 * lh * rstep;                          /* this lane's element offset */                                \
      if (q0 + 32 <= p.Lq) {                               /* whole sub-block inside the sequence */                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) unsafeAtomicAdd(rowp + ((r & 3) + 8 * (r >> 2)) * rstep + loff, dqacc[r] * 0.6931471805599453f); \
      } else {                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                              \
          if (q0 + acc_row(r, lh) < p.Lq) unsafeAtomicAdd(rowp + ((r & 3) + 8 * (r >> 2)) * rstep + loff, dqacc[r] * 0.6931471805599453f); \
      }                                                                                             \
    }                                                                                               \
  }
    for (int qt = 0; qt < nqt; ++qt) {
      const int buf = qt & 1;
      if (DROP && (qt & (ATTN_DROP_QWIN / FB_QROWS - 1)) == 0) {      // a new 256-row window of query rows: re-hash this lane's three column keys
#pragma unroll
        for (int kb = 0; kb < FB_KB; ++kb)
          ck2[kb] = attn_drop_colkey16(salt, kp0 + wave * FB_WKEYS + kb * 32 + lr, (qt * FB_QROWS) / ATTN_DROP_QWIN) * 0x10001u;
      }
      // hand-off: how far the predecessor has published this tile's running sum - asked now, looked at at the end of phase A
      unsigned fv = 0;
      if constexpr (HO && !TAIL) fv = __hip_atomic_load(flags_pair + qt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // global loads of the next tile in sequence (past the end: clamped rows, harmless).  Pipelined form: issued a third of the way
      // into phase A instead of here - at the top of the tile the memory pipeline is still draining the 16 atomics per lane of the
      // previous tile, and the loads are not needed before the end of the phase
      if constexpr (!FULL) FB_STAGE_LOAD();
      const char* qb_ = stage + buf * FB_STAGE;
      const char* dob_ = qb_ + FB_TILE;
      const float* lse_s = reinterpret_cast<const float*>(qb_ + 2 * FB_TILE);
      const float* del_s = lse_s + FB_QROWS;
      // ================= phase A: S, dP, dS, dV^T, dK^T per (query sub-block, key block) =================
      if constexpr (FULL) {
        {   // (EDGE: a wave whose 96 keys all lie past the list runs the phase as well - its P is forced to 0; a branch around the
            // phase costs the edge kernel ~100 spilled registers, whose scratch reloads queue behind the atomics, for no gain: the
            // workgroup waits for its busiest wave at the barrier anyway)
        // Software pipeline over the tile's six blocks b_i = (query sub-block i / 3, key block i % 3).  One wave per SIMD issues in
        // order, so an MFMA only overlaps VALU / LDS work that stands BETWEEN it and the next MFMA in the instruction stream.
        // Slots of 8 MFMAs each, fenced into one-MFMA groups:
        //   G1(b0) | G1(b1) + E(b0) | G2(b0) + M(b0) | G1(b2) + E(b1) | G2(b1) + M(b1) | ... | G1(b5) + E(b4) | G2(b4) + M(b4), E(b5) | G2(b5) + M(b5)
        // G1 = S, dP chains (row constants through the C operand of the first MFMA); E = P = exp2(S') and its bf16 operand
        // words (2 v_exp + 1 cvt per group); G2 = 4 dV^T MFMAs (need P) then 4 dK^T MFMAs (need dS); M = dS = P dP' (4 mul + 2 cvt
        // per group, in the dV^T half).  Every consumer stands at least one MFMA group behind the MFMA that produces its input
        // (S3 is the 7th MFMA of a G1, E starts in the next slot; dP3 is the 8th, M starts a whole slot later): the wait states the
        // hardware does not interlock are covered by construction.  LDS loads of a slot's successor are issued at its head.
        bf16x8 qT[2][2], doT[2][2];
        uint32_t mw[8], rkw[8];       // dropout: mask words of the block in its softmax, row-key words of the next one
        f32x4 dl[4];                  // dropout: -delta of the rows of the block whose dS is formed
        const uint32_t* rk_s = reinterpret_cast<const uint32_t*>(del_s + FB_QROWS);
#define FB_LD_RK(i_)     /* row-key words of block i: registers (2m, 2m+1) = rows 8g + 4lh + {0,1} / {2,3}: word (sb*32 + 8g + 4lh)/2 + (m & 1) */ \
  if (DROP) { _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                       \
    const uint2 w2 = *reinterpret_cast<const uint2*>(rk_s + ((i_) / 3) * 16 + 4 * g + 2 * lh);      \
    rkw[2 * g] = w2.x; rkw[2 * g + 1] = w2.y; } }
#define FB_LD_DL(i_)                                                                                \
  if (DROP) { _Pragma("unroll") for (int g = 0; g < 4; ++g) dl[g] = *reinterpret_cast<const f32x4*>(del_s + ((i_) / 3) * 32 + 8 * g + 4 * lh); }
        int thr[2] = {0, 0};          // EDGE: visibility threshold of the block whose softmax runs (by block parity)
        const int rowb = qt * FB_QROWS + 4 * lh;
#define FB_THR(i_) if (EDGE) thr[(i_) & 1] = qmin[(i_) % 3] - rowb - ((i_) / 3) * 32;
        uint32_t pfw[8], dsw[8];
        f32x2 dst[2];                 // the chunk pair of dS in flight between two MFMA groups
        // 8-byte stores of rows lr and lr + 1 would hit the same LDS banks (the chunk swizzle ignores bit 0 of the row, and an
        // 8-byte store spans half a chunk): odd rows take the other half of the chunk - 10 % of this kernel's LDS cycles were
        // bank conflicts of these stores (profiles/mfma_busy.json, round 2 / 3)
        char* dsw_ = dsimg + wave * (FB_WKEYS * 128) + 8 * (lh ^ (lr & 1));
#define FB_LD_QT(sb_)                                                                               \
  _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                     \
  _Pragma("unroll") for (int db = 0; db < 2; ++db) {                                                \
    qT[s][db] = fb_tr(qb_ + ((sb_) * 32 + 16 * s) * 128, va[db]);                                   \
    doT[s][db] = fb_tr(dob_ + ((sb_) * 32 + 16 * s) * 128, va[db]);                                 \
  }
#define FB_ST_DS(i_)     /* dS^T image rows of block i: queries sb*32 + 16s + {0..3, 8..11} + 4lh of this lane's key */  \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                   \
    *reinterpret_cast<uint2*>(dsw_ + ((i_) % 3) * 4096 + (wrow + (((4 * ((i_) / 3) + 2 * s) << 4) ^ wxor))) = make_uint2(dsw[4 * s], dsw[4 * s + 1]);          \
    *reinterpret_cast<uint2*>(dsw_ + ((i_) % 3) * 4096 + (wrow + (((4 * ((i_) / 3) + 2 * s + 1) << 4) ^ wxor))) = make_uint2(dsw[4 * s + 2], dsw[4 * s + 3]);  \
  }
#define FB_G1(i_, m_) fb_g1<i_, m_, DROP>(sacc, dpacc, qf, dof, kf, vf)
#define FB_G2(i_, m_) fb_g2<i_, m_>(dvacc, dkacc, doT, qT, pfw, dsw)
#define FB_E(i_, m_) fb_ve<i_, m_, EDGE, DROP>(sacc, pfw, thr[(i_) & 1], mw, rkw, ck2[(i_) % 3], th2)
#define FB_M2(i_, m_) fb_vm2<i_, m_, DROP, true, true>(sacc, dpacc, dsw, mw, dl, drop_inv, dst)
#define FB_M2H(i_, m_) fb_vm2<i_, m_, DROP, true, false>(sacc, dpacc, dsw, mw, dl, drop_inv, dst)
#define FB_M2T(i_, m_) fb_vm2<i_, m_, DROP, false, true>(sacc, dpacc, dsw, mw, dl, drop_inv, dst)
        // slot "G1(n) + E(e)": eight groups of one MFMA of G1(b_n) and one chunk of E(b_e)
#define FB_SLOT_G1E(n_, e_)                                                                         \
  FB_THR(e_); FB_LD_RK(e_);                                                                         \
  FB_G1(n_, 0); FB_E(e_, 0); FB_FENCE(); FB_G1(n_, 1); FB_E(e_, 1); FB_FENCE(); FB_G1(n_, 2); FB_E(e_, 2); FB_FENCE();            \
  FB_G1(n_, 3); FB_E(e_, 3); FB_FENCE(); FB_G1(n_, 4); FB_E(e_, 4); FB_FENCE(); FB_G1(n_, 5); FB_E(e_, 5); FB_FENCE();            \
  FB_G1(n_, 6); FB_E(e_, 6); FB_FENCE(); FB_G1(n_, 7); FB_E(e_, 7); FB_FENCE();
        // slot "G2(i) + M(i)": dV^T MFMAs with two chunks of M each, then the dK^T MFMAs beside the dS^T stores
#define FB_SLOT_G2M(i_) FB_SLOT_G2M_(i_, true)
#define FB_SLOT_G2M_(i_, st_)                                                                       \
  FB_LD_DL(i_);                                                                                     \
  FB_G2(i_, 0); FB_M2(i_, 0); FB_FENCE(); FB_G2(i_, 1); FB_M2(i_, 2); FB_FENCE();                                                \
  FB_G2(i_, 2); FB_M2H(i_, 4); FB_FENCE(); FB_G2(i_, 3); FB_M2T(i_, 4); FB_FENCE();                                               \
  /* the dK^T MFMAs of s = 0 need chunks 0..3 only: chunks 6, 7 of dS are formed beside the first of them (an MFMA group ahead of */   \
  /* the s = 1 MFMAs that read them: the asm MFMAs get no hazard nops), the seeds of block i + 2 are fetched once the last chunk */     \
  /* has read this block's accumulators, the dS^T stores follow */                                                                   \
  FB_G2(i_, 4); FB_M2(i_, 6); FB_FENCE(); FB_G2(i_, 5); if ((i_) + 2 < 6) { FB_LD_SEEDS(lse_s, del_s, (i_) + 2); } FB_FENCE(); \
  FB_G2(i_, 6); if (st_) { FB_ST_DS(i_); } FB_FENCE(); FB_G2(i_, 7); FB_FENCE();
        if (!PREF) { FB_LD_QF(qb_, dob_, 0); FB_LD_SEEDS(lse_s, del_s, 0); FB_LD_KF(0); }
        FB_FENCE();
        // slot 0: G1(b0), no VALU work of this tile to pair yet; the transposed fragments of sub-block 0 arrive meanwhile.  ILV: twelve of
        // the 24 dQ MFMAs of the PREVIOUS tile ride here and twelve in slot 1 (tile 0 has no predecessor: they run on whatever the image
        // holds and the result is dropped - no branch in the slots)
        if constexpr (ILV) {
#define FB_DQ_STEP(k_)                                                                              \
  if ((k_) == 0) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(dqacc) : "v"(FB_U4(afA[0])), "v"(FB_U4(bfA[0])));           \
  else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dqacc) : "v"(FB_U4(afA[(k_) % FB_DQ_DEPTH])), "v"(FB_U4(bfA[(k_) % FB_DQ_DEPTH])));        \
  if ((k_) + FB_DQ_DEPTH < FB_KEYS / 16) {                                                                    \
    afA[(k_) % FB_DQ_DEPTH] = fb_tr_abs(vaq_abs, (16 * ((k_) + FB_DQ_DEPTH)) * 128);                                    \
    bfA[(k_) % FB_DQ_DEPTH] = fb_tr_abs(vad_abs, (16 * ((k_) + FB_DQ_DEPTH)) * 128);                                    \
  }                                                                                                 \
  FB_FENCE();
          FB_ABS_HERE();
          FB_G1(0, 0); FB_LD_SEEDS(lse_s, del_s, 1); FB_FENCE(); FB_DQ_STEP(0); FB_DQ_STEP(1);
          FB_G1(0, 1); FB_FENCE(); FB_DQ_STEP(2); FB_DQ_STEP(3);
          FB_G1(0, 2); FB_FENCE(); FB_DQ_STEP(4); FB_DQ_STEP(5);
          FB_G1(0, 3); FB_FENCE(); FB_DQ_STEP(6); FB_DQ_STEP(7);
          FB_G1(0, 4); FB_FENCE(); FB_DQ_STEP(8);
          FB_G1(0, 5); FB_FENCE(); FB_DQ_STEP(9);
          FB_G1(0, 6); FB_FENCE(); FB_DQ_STEP(10);
          FB_G1(0, 7); FB_FENCE(); FB_DQ_STEP(11);
          FB_LD_KF(1); FB_FENCE();
          FB_THR(0); FB_LD_RK(0);
          FB_G1(1, 0); FB_E(0, 0); FB_FENCE(); FB_DQ_STEP(12); FB_DQ_STEP(13);
          FB_G1(1, 1); FB_E(0, 1); FB_FENCE(); FB_DQ_STEP(14); FB_DQ_STEP(15);
          FB_G1(1, 2); FB_E(0, 2); FB_FENCE(); FB_DQ_STEP(16); FB_DQ_STEP(17);
          FB_G1(1, 3); FB_E(0, 3); FB_FENCE(); FB_DQ_STEP(18); FB_DQ_STEP(19);
          FB_G1(1, 4); FB_E(0, 4); FB_FENCE(); FB_DQ_STEP(20);
          FB_G1(1, 5); FB_E(0, 5); FB_FENCE(); FB_DQ_STEP(21);
          FB_G1(1, 6); FB_E(0, 6); FB_FENCE(); FB_DQ_STEP(22);
          FB_G1(1, 7); FB_E(0, 7); FB_FENCE(); FB_DQ_STEP(23);
          // The last dQ MFMA writes its 16 registers at the END of its 32 cycles, and the compiler - which does not see an MFMA in the asm
          // statement - hands those registers to whatever comes next: on tile 0, where the result is dropped, the exponentials of slot 2
          // landed in them and were overwritten by the MFMA's late write (garbage P for one key block, seen as 1e37 in dK / dV with the
          // atomic form and dropout).  Twelve wait states on EVERY path, not only inside the hand-off below.
          asm volatile("s_nop 11" : "+v"(dqacc));
          // dQ of the previous tile leaves: running sum + this block's share -> the hand-off buffer (or the atomics).  Hand-off form: NO
          // branch around it on tile 0 - its byte offset ((-1) * 4 + wave) * 4096 wraps far behind the descriptor's records and the four
          // stores are dropped by the range check; a branch here splits the slot's basic block, and the compiler then SINKS the
          // exponentials of E(b0) out of the MFMA shadow into the block behind the branch (seen in the ISA)
          if constexpr (HO) { FB_DQ_FINALIZE(qt - 1); }
          else if (qt > 0) { FB_DQ_FINALIZE(qt - 1); }
        } else {
        FB_G1(0, 0); FB_LD_QT(0); FB_LD_SEEDS(lse_s, del_s, 1); FB_FENCE(); FB_G1(0, 1); FB_FENCE(); FB_G1(0, 2); FB_FENCE(); FB_G1(0, 3); FB_FENCE();
        FB_G1(0, 4); FB_FENCE(); FB_G1(0, 5); FB_FENCE(); FB_G1(0, 6); FB_FENCE(); FB_G1(0, 7); FB_FENCE();
        FB_LD_KF(1); FB_FENCE();
        FB_SLOT_G1E(1, 0);
        }
        // the barrier that ends the PREVIOUS tile (every wave is done reading its dS^T image) stands here, ahead of the first dS^T
        // store of this tile, instead of behind the atomics: a wave that got its atomics out early starts the next tile.  A bare
        // s_barrier: it orders later LDS writes behind earlier LDS reads whose data has long been consumed by MFMAs - nothing
        // to wait for (__syncthreads would drain the LDS loads in flight here)
        constexpr bool SOFT = ILV && FB_SOFT_SYNC;
        if constexpr (SOFT) {
          // this wave's reads of the dS^T image are done (their data went into MFMAs that have issued; LDS operations of a wave execute in order)
          if (lane == 0) __hip_atomic_fetch_add(rd_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
#if !(FB_ABL_NOBAR & 1)
          asm volatile("s_barrier" ::: "memory");
#endif
        }
        if constexpr (ILV) { FB_LD_QT(0); }                  // (ILV: the transposed Q / dO fragments of sub-block 0, first used right below)
        FB_LD_KF(2); FB_FENCE();
        if constexpr (SOFT) {
          FB_SLOT_G2M_(0, false);
          // every wave's slot 1 of this tile is through: the image may be written.  (Between two slots: a branch here splits no slot.)
          const unsigned rd_target = 4u * (unsigned)(qt + 1);
          while (__hip_atomic_load(rd_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < rd_target) __builtin_amdgcn_s_sleep(1);
          FB_ST_DS(0); FB_FENCE();
        } else {
          FB_SLOT_G2M(0);
        }
        FB_SLOT_G1E(2, 1);                                   // last use of sub-block 0's row fragments and row constants
        FB_LD_QF(qb_, dob_, 1); FB_LD_KF(0); FB_FENCE();
        FB_SLOT_G2M(1);
        FB_STAGE_LOAD(); FB_FENCE();
        FB_SLOT_G1E(3, 2);
        FB_LD_KF(1); FB_FENCE();
        FB_SLOT_G2M(2);                                      // last use of sub-block 0's transposed fragments
        FB_LD_QT(1); FB_FENCE();
        FB_SLOT_G1E(4, 3);
        FB_LD_KF(2); FB_FENCE();
        FB_SLOT_G2M(3);
        FB_SLOT_G1E(5, 4);
        // slot "G2(b4) + M(b4) + E(b5)": E(b5) only behind the dV^T MFMAs of b4, which still read the operand words of P(b4)
        FB_G2(4, 0); FB_M2(4, 0); FB_FENCE(); FB_G2(4, 1); FB_M2(4, 2); FB_FENCE();
        FB_G2(4, 2); FB_M2(4, 4); FB_FENCE(); FB_G2(4, 3); FB_M2(4, 6); FB_FENCE();
        FB_THR(5); FB_LD_RK(5);
        FB_G2(4, 4); FB_ST_DS(4); FB_E(5, 0); FB_E(5, 1); FB_FENCE(); FB_G2(4, 5); FB_E(5, 2); FB_E(5, 3); FB_FENCE();
        FB_G2(4, 6); FB_E(5, 4); FB_E(5, 5); FB_FENCE(); FB_G2(4, 7); FB_E(5, 6); FB_E(5, 7); FB_FENCE();
        FB_SLOT_G2M(5);
#undef FB_THR
#undef FB_LD_RK
#undef FB_LD_DL
#undef FB_LD_QT
#undef FB_ST_DS
#undef FB_G1
#undef FB_G2
#undef FB_E
#undef FB_M2
#undef FB_M2H
#undef FB_M2T
#undef FB_SLOT_G1E
#undef FB_SLOT_G2M
#undef FB_SLOT_G2M_
        }
      } else {
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        bf16x8 qf[4], dof[4], qT[2][2], doT[2][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          qf[s] = lds_row_frag(qb_, sb * 32 + lr, s, lh);
          dof[s] = lds_row_frag(dob_, sb * 32 + lr, s, lh);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            qT[s][db] = lds_tr_frag(qb_, sb * 32 + 16 * s, db, lane);
            doT[s][db] = lds_tr_frag(dob_, sb * 32 + 16 * s, db, lane);
          }
#pragma unroll
        for (int kb = 0; kb < FB_KB; ++kb) {
          const int keyrow0 = wave * FB_WKEYS + kb * 32;
          if (FULL || keyrow0 < nkeys_wg) {                  // wave-uniform: key blocks past the list are skipped
            // accumulators start from the row constants of this lane's rows acc_row(r, lh) = 8g + 4lh + j (broadcast LDS reads)
            f32x16 sacc, dpacc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + sb * 32 + 8 * g + 4 * lh);
              const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + sb * 32 + 8 * g + 4 * lh);
#pragma unroll
              for (int j = 0; j < 4; ++j) { sacc[4 * g + j] = l4[j]; dpacc[4 * g + j] = d4[j]; }
            }
            bf16x8 kf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) kf[s] = lds_row_frag(kimg, keyrow0 + lr, s, lh);
            fb_mfma_sdp(sacc, dpacc, qf, kf, dof, vf[kb]);           // c*S[q, key] - LSE*log2e ; dP[q, key] - delta
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = fast_exp2(sacc[r]);
            if (!FULL && edge_wg) {          // decoder / validity rule behind a real uniform branch (see attn_dkdv_bf16_sweep.inc)
              asm volatile("" ::: "memory");
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int qdec = qt * FB_QROWS + sb * 32 + acc_row(r, lh) - p.dec_q0;
                const bool ok = kvalid[kb] && (kdec[kb] < 0 || qdec >= kdec[kb]);
                sacc[r] = ok ? sacc[r] : 0.f;
              }
            }
            f32x16 pdrop = sacc;                             // P as dV sees it (dropped entries cleared)
            if (DROP) {      // plain per-score form (this sweep only runs for key blocks beyond the static bound)
              const uint32_t ck16 = ck2[kb] & 0xFFFFu;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                int qg = qt * FB_QROWS + sb * 32 + acc_row(r, lh);
                qg = qg < p.Lq ? qg : p.Lq - 1;
                const bool keep = attn_drop_keep16(attn_drop_rowkey16(salt, qg, kbw), ck16, p.drop_thresh);
                // the chain was seeded with -delta: dP' = dP - delta;  dS = P (keep ? dP / (1-p) : 0) - P delta
                const float nd = del_s[sb * 32 + acc_row(r, lh)];
                dpacc[r] = sacc[r] * ((keep ? (dpacc[r] - nd) * drop_inv : 0.f) + nd);
                pdrop[r] = keep ? sacc[r] : 0.f;
              }
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) dpacc[r] = sacc[r] * dpacc[r];
            }
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) { pf[s] = acc_to_frag(pdrop, s); dsf[s] = acc_to_frag(dpacc, s); }
            fb_mfma_dvdk(dvacc[kb][0], dvacc[kb][1], dkacc[kb][0], dkacc[kb][1], doT, qT, pf, dsf);
            // dS^T image: this lane's key row, queries sb*32 + 16s + {0..3, 8..11} + 4lh: two 8-byte stores per s
            char* dsrow = dsimg + 8 * (lh ^ (lr & 1));        // (keyrow0 is a multiple of 32: the row's parity is lr's)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              const uint4 w = __builtin_bit_cast(uint4, dsf[s]);
              *reinterpret_cast<uint2*>(dsrow + tile_off(keyrow0 + lr, 4 * sb + 2 * s)) = make_uint2(w.x, w.y);
              *reinterpret_cast<uint2*>(dsrow + tile_off(keyrow0 + lr, 4 * sb + 2 * s + 1)) = make_uint2(w.z, w.w);
            }
          }
        }
      }
      }   // general form
      // Stage the next tile here, at the end of phase A: its DMA pieces, issued a third of the way into the phase, have long landed; the
      // buffer was last read in phase A of the previous tile, and the barrier below publishes it.
      // K^T fragments of the first steps of the dQ product: they do not depend on this tile, so they are fetched ahead of the barrier
      // (their registers were the transposed Q / dO fragments until a moment ago)
      if constexpr (ILV) {
        FB_ABS_HERE();
#pragma unroll
        for (int u = 0; u < FB_DQ_DEPTH; ++u) bfA[u] = fb_tr_abs(vad_abs, (16 * u) * 128);
      }
      if constexpr (HO) {
        // every wave: its running-sum stores of the PREVIOUS tile have landed (the flag below is stored behind this wait and the
        // barrier: Guideline 16 R1); the stage loads of this tile are younger and are needed right below anyway
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!TAIL) FB_FLAG_WAIT(flags_pair + qt, fv);          // the predecessor has not published that tile yet: bounded spin
      }
      FB_STAGE_WRITE(buf ^ 1);
      if constexpr (HO) {
        // hand-off: the running sum of the blocks before this one (zeros for block 0: zero-record descriptor).  This wave's own poll has
        // matched: its loads of the sum may go out now (every load of handed-off bytes is an sc1 load issued by a wave behind its own
        // matching poll) - ahead of the barrier, behind the staging wait (which would wait for them too); they have the barrier and the
        // dQ MFMAs of this tile (ILV: slot 0 of the next tile) to come back
        const unsigned ho_ld = (unsigned)((qt * 4 + wave_u) * 4096 + lane * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) pin[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_ld, ho_ld + g * 1024, 0, 16 /* sc1 */);
      }
#if !(FB_ABL_NOBAR & 2)
      __syncthreads();                                       // the dS^T image of this query tile is complete
#else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
      if constexpr (HO && !TAIL) {
        // publish the previous tile's running sum: every wave drained its stores before the barrier above
        if ((!CAN_LAST || !ho_last) && tid == 0 && qt > 0 && !w.never_publish)
          __hip_atomic_store(flags_pair + (qt - 1), (unsigned)(kbw + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if constexpr (ILV) {
        // the dQ product of this tile runs in slots 0 and 1 of the next one (or in the drain behind the loop): its first dS^T fragments and
        // the next tile's first operands (its stage buffer was published by the barrier above) are fetched here
        FB_ABS_HERE();
#pragma unroll
        for (int k = 0; k < FB_DQ_DEPTH; ++k) afA[k] = fb_tr_abs(vaq_abs, (16 * k) * 128);
        const char* nq_ = stage + (buf ^ 1) * FB_STAGE;
        FB_LD_QF(nq_, nq_ + FB_TILE, 0);
        FB_LD_SEEDS(reinterpret_cast<const float*>(nq_ + 2 * FB_TILE), reinterpret_cast<const float*>(nq_ + 2 * FB_TILE) + FB_QROWS, 0);
        FB_LD_KF(0);
        FB_FENCE();
      } else {
      // ================= phase B (serial form): dQ[32 q, 32 d] of this wave over all keys of the workgroup =================
#pragma unroll
        for (int i = 0; i < 16; ++i) dqacc[i] = 0.f;
        // steps of 16 keys in groups of 4, the steps that hold valid keys (rows past them may never have been written)
#define FB_DQ_LOAD_A(af_, g_) _Pragma("unroll") for (int u = 0; u < 4; ++u) af_[u] = fb_tr_abs(vaq_abs, (16 * (4 * (g_) + u)) * 128);
#define FB_DQ_LOAD_B(bf_, g_) _Pragma("unroll") for (int u = 0; u < 4; ++u) bf_[u] = fb_tr_abs(vad_abs, (16 * (4 * (g_) + u)) * 128);
#define FB_DQ_LOAD(af_, bf_, g_) FB_DQ_LOAD_A(af_, g_) FB_DQ_LOAD_B(bf_, g_)
#define FB_DQ_MFMA(af_, bf_)                                                                        \
  asm("s_nop 1\n\t"                                                                                \
      "v_mfma_f32_32x32x16_bf16 %0, %1, %5, %0\n\t"                                                \
      "v_mfma_f32_32x32x16_bf16 %0, %2, %6, %0\n\t"                                                \
      "v_mfma_f32_32x32x16_bf16 %0, %3, %7, %0\n\t"                                                \
      "v_mfma_f32_32x32x16_bf16 %0, %4, %8, %0"                                                     \
      : "+v"(dqacc)                                                                                 \
      : "v"(FB_U4(af_[0])), "v"(FB_U4(af_[1])), "v"(FB_U4(af_[2])), "v"(FB_U4(af_[3])), "v"(FB_U4(bf_[0])), "v"(FB_U4(bf_[1])),  \
        "v"(FB_U4(bf_[2])), "v"(FB_U4(bf_[3])));
        {
          const int nsteps = FULL ? (EDGE ? nks : FB_KEYS / 16) : nks;
          int k4 = 0;
          for (; k4 + 4 <= nsteps; k4 += 4) {
            bf16x8 af[4], bfr[4];
            FB_DQ_LOAD(af, bfr, k4 >> 2);
            FB_DQ_MFMA(af, bfr);
          }
          for (; k4 < nsteps; ++k4) {
            const bf16x8 a1 = fb_tr_abs(vaq_abs, (16 * k4) * 128), b1 = fb_tr_abs(vad_abs, (16 * k4) * 128);
            asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dqacc) : "v"(FB_U4(a1)), "v"(FB_U4(b1)));
          }
        }
#undef FB_DQ_LOAD
#undef FB_DQ_LOAD_A
#undef FB_DQ_LOAD_B
#undef FB_DQ_MFMA
        FB_DQ_FINALIZE(qt);
        if constexpr (!FULL) __syncthreads();                // every wave is done reading the dS^T image (pipelined edge sweep: see phase A)
      }
    }
    if constexpr (ILV) {
      // drain: the dQ product of the LAST tile (its dS^T image was completed by the loop's last barrier, its first operands are in
      // registers, its predecessor sum is on its way)
      FB_DQ_STEP(0); FB_DQ_STEP(1); FB_DQ_STEP(2); FB_DQ_STEP(3); FB_DQ_STEP(4); FB_DQ_STEP(5); FB_DQ_STEP(6); FB_DQ_STEP(7);
      FB_DQ_STEP(8); FB_DQ_STEP(9); FB_DQ_STEP(10); FB_DQ_STEP(11); FB_DQ_STEP(12); FB_DQ_STEP(13); FB_DQ_STEP(14); FB_DQ_STEP(15);
      FB_DQ_STEP(16); FB_DQ_STEP(17); FB_DQ_STEP(18); FB_DQ_STEP(19); FB_DQ_STEP(20); FB_DQ_STEP(21); FB_DQ_STEP(22); FB_DQ_STEP(23);
      FB_DQ_FINALIZE(nqt - 1);
    }
    if constexpr (HO) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last tile's stores (the tail launch reads them back in its next key block)
      if constexpr (!TAIL) {
        if (!CAN_LAST || !ho_last) {                         // workgroup-uniform
          __syncthreads();
          if (tid == 0 && !w.never_publish) __hip_atomic_store(flags_pair + (nqt - 1), (unsigned)(kbw + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    };
    if constexpr (MODE == 3) {
      if (edge_wg) sweep(std::true_type{}, std::true_type{});
      else sweep(std::true_type{}, std::false_type{});
    } else {
      sweep(std::integral_constant<bool, MODE != 2>{}, std::integral_constant<bool, MODE == 1>{});
    }
#undef FB_LD_QF
#undef FB_LD_SEEDS
#undef FB_LD_KF
#undef FB_DQ_FINALIZE
#undef FB_DQ_STEP
    // dK^T / dV^T were last written by asm MFMAs the compiler does not see as such: cover MFMA result -> v_accvgpr_read
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb)
      asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dkacc[kb][0]), "+a"(dkacc[kb][1]), "+a"(dvacc[kb][0]), "+a"(dvacc[kb][1]));
#undef FB_STAGE_LOAD
#undef FB_STAGE_WRITE
#undef FB_FLAG_WAIT

    // ---- dK / dV of this wave's keys
#pragma unroll
    for (int kb = 0; kb < FB_KB; ++kb)
      if (kvalid[kb]) {
        bf16_t* dkp = reinterpret_cast<bf16_t*>(p.dk) + (int64_t)b * p.kv_bs + h * 64 + krow[kb] * p.kv_rs;
        bf16_t* dvp = reinterpret_cast<bf16_t*>(p.dv) + (int64_t)b * p.kv_bs + h * 64 + krow[kb] * p.kv_rs;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d = db * 32 + 8 * g + 4 * lh;
            bf16x4 k4 = {(bf16_t)(dkacc[kb][db][4 * g] * p.scale), (bf16_t)(dkacc[kb][db][4 * g + 1] * p.scale),
                         (bf16_t)(dkacc[kb][db][4 * g + 2] * p.scale), (bf16_t)(dkacc[kb][db][4 * g + 3] * p.scale)};
            const float vs_ = DROP ? drop_inv : 1.f;
            bf16x4 v4 = {(bf16_t)(dvacc[kb][db][4 * g] * vs_), (bf16_t)(dvacc[kb][db][4 * g + 1] * vs_), (bf16_t)(dvacc[kb][db][4 * g + 2] * vs_),
                         (bf16_t)(dvacc[kb][db][4 * g + 3] * vs_)};
            *reinterpret_cast<bf16x4*>(dkp + d) = k4;
            *reinterpret_cast<bf16x4*>(dvp + d) = v4;
          }
      }
    if (TAIL) __syncthreads();                               // the next key block rewrites the K image
  } while (TAIL && (++kbw) * FB_KEYS < nk);
}

// delta[b, h, q] = sum_d dO * O (one wave per token row, as attn_delta_kernel), plus the housekeeping of the fused backward in
// the same pass over the rows: zero the row of the fp32 dQ accumulation buffer, and write the exact-zero dK / dV slices of
// rows that no key-list entry points at (prefix rows with row_valid == 0; rows behind the prefix other than this call's decoder rows).
__global__ __launch_bounds__(256) void attn_delta_prep_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, float* __restrict__ delta,
                                                              const float* __restrict__ lse, float* __restrict__ nl, float* __restrict__ nd, int nq_pad,
                                                              float* __restrict__ dq32 /* NULL: no accumulation buffer to zero (hand-off form) */, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv,
                                                              const uint8_t* __restrict__ row_valid, int valid_len, int dec_q0, int n_dec, int B, int H, int Lq,
                                                              int64_t o_rs, int64_t o_bs, int64_t kv_rs, int64_t kv_bs,
                                                              unsigned* __restrict__ slots, int groups, const int32_t* __restrict__ kv_cnt, int dense_keys,
                                                              int kblocks, bf16_t* __restrict__ dq_empty /* hand-off: dq, to zero the rows of a sample WITHOUT keys */,
                                                              int64_t q_rs, int64_t q_bs) {
  const int lane = threadIdx.x & 63;
  // the fused sweep's ticket table (FbWork.slots): per XCD group the first slot of each of its (sample, head) pairs = running count of
  // the key blocks that exist (the sample's key count, capped by the launch's static bound), then the group's total
  if (slots && blockIdx.x == 0 && threadIdx.x < T2S_XCDS) {
    unsigned acc = 0;
    unsigned* t = slots + threadIdx.x * (groups + 1);
    for (int g = 0; g < groups; ++g) {
      t[g] = acc;
      const int bh = g * T2S_XCDS + (int)threadIdx.x;
      if (bh < B * H) {
        const int nk = (kv_cnt ? kv_cnt[bh / H] : dense_keys) + n_dec;
        const int nb = (nk + FB_KEYS - 1) / FB_KEYS;
        acc += (unsigned)(nb < kblocks ? nb : kblocks);
      }
    }
    t[groups] = acc;
  }
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * Lq) return;
  const int b = (int)(row / Lq), q = (int)(row % Lq);
  const bf16_t* op = o + (int64_t)b * o_bs + (int64_t)q * o_rs;
  const bf16_t* dp = dout + (int64_t)b * o_bs + (int64_t)q * o_rs;
  const int nchunk = H * 16;              // 4-element chunks per row
  float* zrow = dq32 ? dq32 + row * (int64_t)(H * 64) : nullptr;
  // hand-off form: dQ is written by the LAST key block of a pair; a sample with an empty key list has no block, so its (exactly zero)
  // dQ rows are written here (the atomic form zero-fills its sum buffer anyway)
  const bool empty = dq_empty && ((kv_cnt ? kv_cnt[b] : dense_keys) + n_dec) <= 0;
  const bool fill = row_valid && (q < valid_len ? !row_valid[(int64_t)b * valid_len + q] : (q < dec_q0 || q >= dec_q0 + n_dec));
  for (int c0 = 0; c0 < nchunk; c0 += 64) {
    const int ci = c0 + lane;
    float s = 0.f;
    if (ci < nchunk) {
      const f32x4 a = Vec4<bf16_t>::load(op + ci * 4), d = Vec4<bf16_t>::load(dp + ci * 4);
      s = a[0] * d[0] + a[1] * d[1] + a[2] * d[2] + a[3] * d[3];
      if (zrow) *reinterpret_cast<f32x4*>(zrow + ci * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (empty) *reinterpret_cast<bf16x4*>(dq_empty + (int64_t)b * q_bs + (int64_t)q * q_rs + ci * 4) = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
      if (fill) {
        const bf16x4 z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        *reinterpret_cast<bf16x4*>(dk + (int64_t)b * kv_bs + (int64_t)q * kv_rs + ci * 4) = z;
        *reinterpret_cast<bf16x4*>(dv + (int64_t)b * kv_bs + (int64_t)q * kv_rs + ci * 4) = z;
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if ((lane & 15) == 0 && ci < nchunk) {
      const int64_t bh = (int64_t)b * H + (ci >> 4);
      delta[bh * Lq + q] = s;
      if (nl) {          // the row constants as the fused sweep seeds its accumulators with them (LDS-DMA copies them raw)
        nl[bh * nq_pad + q] = -(lse[bh * Lq + q] * LOG2E);
        nd[bh * nq_pad + q] = -s;
        // rows behind Lq of the padded arrays (< 64 per pair): P = exp2(-inf) = 0 there, and -delta must not be a NaN left in the
        // workspace by an earlier use (0 * NaN) - written by the wave that holds the sample's last row
        if (q == Lq - 1)
          for (int pq = Lq; pq < nq_pad; ++pq) { nl[bh * nq_pad + pq] = -INFINITY; nd[bh * nq_pad + pq] = 0.f; }
      }
    }
  }
}

// dq (bf16, strided rows inside the fused QKV gradient buffer) = bf16(dq32): 8 elements per thread
__global__ __launch_bounds__(256) void attn_dq_cast_kernel(const float* __restrict__ dq32, bf16_t* __restrict__ dq, int64_t rows_per_b, int width,
                                                           int64_t q_rs, int64_t q_bs, int64_t total8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total8) return;
  const int per_row = width >> 3;
  const int64_t row = i / per_row;
  const int c8 = (int)(i - row * per_row);
  const int64_t b = row / rows_per_b, q = row - b * rows_per_b;
  const f32x4 a = *reinterpret_cast<const f32x4*>(dq32 + row * width + c8 * 8);
  const f32x4 c = *reinterpret_cast<const f32x4*>(dq32 + row * width + c8 * 8 + 4);
  bf16x8 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)c[0], (bf16_t)c[1], (bf16_t)c[2], (bf16_t)c[3]};
  *reinterpret_cast<bf16x8*>(dq + b * q_bs + q * q_rs + c8 * 8) = o;
}

}  // namespace

// Workspace of the fused backward, in bytes, for (B, H, Lq): control words + flags + the fp32 dQ sums (either form)
size_t attn_bwd_fused_workspace_bytes(int B, int H, int Lq) {
  const size_t nqt = ((size_t)Lq + FB_QROWS - 1) / FB_QROWS;
  const size_t groups = ((size_t)B * H + T2S_XCDS - 1) / T2S_XCDS;
  const size_t ctrl = ((size_t)FB_CTRL_WORDS * 4 + (size_t)B * H * nqt * 4 + T2S_XCDS * (groups + 1) * 4 + 255) / 256 * 256;
  const size_t rowc = 2 * (size_t)B * H * nqt * FB_QROWS * 4;            // -lse log2e and -delta per padded query row (LDS-DMA sources)
  const size_t sums = (size_t)B * H * nqt * (FB_QROWS * 64 * 4);        // >= B * Lq * H * 64 * 4, the atomic form's buffer
  return ctrl + rowc + sums;
}

// Fused backward (bf16): delta + housekeeping, the 5-product kernel (+ its tail launch, see attn_dkdv_bf16.hip); dQ across key
// blocks by the ordered hand-off (handoff != 0: no zero fill, no cast pass, bit-reproducible) or by fp32 atomics + the cast.
int launch_attn_bwd_fused_bf16(const AttnParams& p_in, int max_keys, void* workspace, size_t workspace_bytes, int dq_mode, hipStream_t st) {
  AttnParams p = p_in;
  const int handoff = dq_mode & 0xff;
  const bool diag_dead = handoff && (dq_mode & 0x100);      // tests: the hand-off with a dead predecessor (spin limit 0, flags never published)
  const bool agent_scope = handoff && (dq_mode & 0x200);    // write-through running sums (a device whose XCD groups are not XCD-local)
  const bool diag_misplaced = handoff && (dq_mode & 0x400); // tests: the placement check sees two XCDs in every group
  if (workspace_bytes < attn_bwd_fused_workspace_bytes(p.B, p.H, p.Lq)) {
    t2s_set_error("attn_bwd_fused: workspace of %zu bytes, %zu needed (t2s_attn_bwd_fused_workspace_bytes)", workspace_bytes,
                  attn_bwd_fused_workspace_bytes(p.B, p.H, p.Lq));
    return 2;
  }
  const size_t nqt = ((size_t)p.Lq + FB_QROWS - 1) / FB_QROWS;
  const size_t groups = ((size_t)p.B * p.H + T2S_XCDS - 1) / T2S_XCDS;
  const size_t ctrl = ((size_t)FB_CTRL_WORDS * 4 + (size_t)p.B * p.H * nqt * 4 + T2S_XCDS * (groups + 1) * 4 + 255) / 256 * 256;
  FbWork w;
  w.tickets = reinterpret_cast<unsigned*>(workspace);
  w.status = w.tickets + 3 * T2S_XCDS;
  w.flags = w.tickets + FB_CTRL_WORDS;
  unsigned* const slots = w.flags + (size_t)p.B * p.H * nqt;          // behind the flags, inside the control block
  w.slots = slots;
  w.groups = (int)groups;
  const size_t rowc_n = (size_t)p.B * p.H * nqt * FB_QROWS;               // elements of each row-constant array
  float* const nl = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ctrl);
  float* const nd = nl + rowc_n;
  w.nl = nl;
  w.nd = nd;
  w.part = nd + rowc_n;
  w.handoff = handoff;
  w.spin_limit = diag_dead ? 0u : FB_SPIN_LIMIT;
  w.never_publish = diag_dead ? 1 : 0;
  w.diag_misplaced = diag_misplaced ? 1 : 0;
  float* const dq32 = w.part;
  // > 64 KB of LDS per workgroup needs the opt-in: once per device (a flag per device ordinal is the only state kept)
  static bool lds_reserved[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev < 0 || !lds_reserved[dev]) {
#define FB_K(I_, M_, D_) reinterpret_cast<const void*>(&attn_bwd_fused_bf16_kernel<I_, M_, D_, false>), reinterpret_cast<const void*>(&attn_bwd_fused_bf16_kernel<I_, M_, D_, true>)
#define FB_KW(I_, M_, D_) reinterpret_cast<const void*>(&attn_bwd_fused_bf16_kernel<I_, M_, D_, true, true>)      /* write-through sums: MODE 3 and the tail launch only */
  const void* kernels[] = {FB_K(true, 0, false), FB_K(true, 1, false), FB_K(true, 2, false), FB_K(false, 0, false), FB_K(false, 1, false),
                           FB_K(true, 0, true),  FB_K(true, 1, true),  FB_K(true, 2, true),  FB_K(false, 0, true),  FB_K(false, 1, true),
                           FB_K(true, 3, false), FB_K(false, 3, false), FB_K(true, 3, true), FB_K(false, 3, true),
                           FB_KW(true, 3, false), FB_KW(false, 3, false), FB_KW(true, 3, true), FB_KW(false, 3, true), FB_KW(true, 2, false), FB_KW(true, 2, true)};
#undef FB_KW
#undef FB_K
  for (const void* k : kernels)
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, FB_SMEM) != hipSuccess) {
      t2s_set_error("attn_bwd_fused: cannot reserve %d bytes of LDS per workgroup", FB_SMEM);
      return 3;
    }
  if (dev >= 0) lds_reserved[dev] = true;
  }
  {
    // tickets, status and flags: zeroed on every call (a replayed or repeated launch starts from a clean protocol state); the
    // atomic form only needs a clean status word
    if (hipMemsetAsync(workspace, 0, handoff ? ctrl : (size_t)FB_CTRL_WORDS * 4, st) != hipSuccess) {
      t2s_set_error("attn_bwd_fused: cannot clear the hand-off control block");
      return 3;
    }
  }
  const int64_t rows = (int64_t)p.B * p.Lq;
  p.kblocks = (max_keys + FB_KEYS - 1) / FB_KEYS;
  hipLaunchKernelGGL(attn_delta_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)p.o, (const bf16_t*)p.dout, p.delta,
                     (const float*)p.lse, nl, nd, (int)(nqt * FB_QROWS), handoff ? nullptr : dq32, (bf16_t*)p.dk, (bf16_t*)p.dv, p.row_valid, p.valid_len, p.dec_q0, p.n_dec, p.B, p.H, p.Lq, p.o_rs, p.o_bs,
                     p.kv_rs, p.kv_bs, handoff ? slots : nullptr, (int)groups, p.kv_idx ? p.kv_cnt : nullptr, p.idx_cap - p.n_dec, p.kblocks,
                     handoff ? (bf16_t*)p.dq : nullptr, p.q_rs, p.q_bs);
  T2S_CHECK_LAUNCH("attn_bwd_fused (delta prep)");
  dim3 grid(attn_xcd_grid(p.kblocks, p.H, p.B)), block(256), tail(attn_xcd_grid(1, p.H, p.B));
#define FB_LAUNCH2(IDX_, MODE_, DROP_, grid_)                                                                            \
  if (handoff) hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<IDX_, MODE_, DROP_, true>), grid_, block, FB_SMEM, st, p, w);   \
  else hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<IDX_, MODE_, DROP_, false>), grid_, block, FB_SMEM, st, p, w);
  // the shipped launch forms (MODE 3, tail) also exist with write-through sums
#define FB_LAUNCH2W(IDX_, MODE_, DROP_, grid_)                                                                           \
  if (handoff && agent_scope) hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<IDX_, MODE_, DROP_, true, true>), grid_, block, FB_SMEM, st, p, w); \
  else { FB_LAUNCH2(IDX_, MODE_, DROP_, grid_) }
#define FB_LAUNCHW(IDX_, MODE_, grid_)                                                                                 \
  if (p.drop_thresh) { FB_LAUNCH2W(IDX_, MODE_, true, grid_) } else { FB_LAUNCH2W(IDX_, MODE_, false, grid_) }          \
  T2S_CHECK_LAUNCH("attn_bwd_fused (five-product kernel)");
#define FB_LAUNCH(IDX_, MODE_, grid_)                                                                                  \
  if (p.drop_thresh) { FB_LAUNCH2(IDX_, MODE_, true, grid_) } else { FB_LAUNCH2(IDX_, MODE_, false, grid_) }            \
  T2S_CHECK_LAUNCH("attn_bwd_fused (five-product kernel)");
  const char* split_env = getenv("T2S_FB_SPLIT_EDGE");          // A/B runs: the two-launch form of rounds 2-3 (read per call)
  const bool split_edge = split_env && split_env[0] == '1' && !agent_scope;      // (the write-through form exists as one launch only)
  if (p.kv_idx) {
    if (split_edge) {
      FB_LAUNCH(true, 0, grid);
      FB_LAUNCH(true, 1, grid);
    } else {
      FB_LAUNCHW(true, 3, grid);
    }
    if (p.kblocks * FB_KEYS < p.idx_cap) { FB_LAUNCHW(true, 2, tail); }
  } else {
    if (split_edge) {
      FB_LAUNCH(false, 0, grid);
      FB_LAUNCH(false, 1, grid);
    } else {
      FB_LAUNCHW(false, 3, grid);
    }
  }
#undef FB_LAUNCHW
#undef FB_LAUNCH2W
#undef FB_LAUNCH
#undef FB_LAUNCH2
  if (!handoff) {
    const int width = p.H * 64;
    const int64_t total8 = rows * (width / 8);
    hipLaunchKernelGGL(attn_dq_cast_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, st, dq32, (bf16_t*)p.dq, (int64_t)p.Lq, width,
                       p.q_rs, p.q_bs, total8);
    T2S_CHECK_LAUNCH("attn_bwd_fused (dQ cast)");
  }
  return 0;
}
