// Own bf16 MFMA GEMM family for gfx950 (round 5): the linear layers of the third-party BERT block
// the reference calls at pythia/models/t2s.py:423-427,538-542,622-626 (BertSelfOutput / BertIntermediate / BertOutput dense layers) and
// their gradients (autograd of the same calls, stepped by pythia/trainers/base_trainer.py:262-272), with the epilogues a library GEMM
// cannot fuse.  bf16 operands, fp32 accumulation.
//
//   NT   C[M, N]  = A[M, K] W[N, K]^T (+ bias)            forward and input-gradient GEMMs (the weight or its transposed copy is [N, K])
//        epilogues: bf16 store | bf16 accumulate (C += ..., the residual branch of a LayerNorm backward) |
//                   du = (A W^T) * gelu'(u) in fp32, one rounding, + per-row-block column sums (the FFN bias gradient): replaces the
//                   standalone gelu_bwd pass | u = A W^T + bias and g = gelu(u) both stored (replaces the standalone gelu_fwd pass)
//   TN   dW[No, Ni] = dY[rows, No]^T X[rows, Ni]            weight gradients: the contraction runs over the 650 k token rows into a tiny
//        output; split over row groups (one 256 x 256 tile x one row group per workgroup: 252 workgroups for the FFN weights), fp32 slabs
//        summed in a fixed order by a second kernel - bit-reproducible.
//
// Shape of the main loop - cdna_hip_programming.md section 5, "The 256^2 8-phase template" (its example file is not in this image; the
// schedule below is re-derived from its rules and every hazard argued in place):
//   * tile 256 x 256 x 64, 512 threads = 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave = 128 fp32 accumulators per lane,
//     v_mfma_f32_16x16x32_bf16 (64 per wave and K-tile); the accumulator tile is C^T (MFMA A operand = the N side): a lane owns 4
//     consecutive output columns of one row per register group;
//   * LDS: 2 K-tile buffers x 4 HALF-TILES (A0 A1 B0 B1, 16 KB each = one operand's rows (m-half h of every wave row / n-half h of every
//     wave column) x 64 k) = 128 KB.  A half-tile is filled by ONE LDS-DMA instruction pair per wave (buffer_load_dwordx4 ... lds, 1 KB per
//     wave-instruction); the bank swizzle is on the per-lane SOURCE address (rule 21), the image is lane-linear;
//   * a K-tile = 4 phases, one C quadrant (64 x 32 per wave, 16 MFMAs) each, order (m0,n0) (m0,n1) (m1,n1) (m1,n0): fragment reads
//     B0+A0 (12 ds_read_b128) | B1 (4) | A1 (8) | none (B0 stays in registers);
//     phase = { fragment reads of this phase; DMA of one half-tile; [counted waits]; s_barrier; lgkmcnt(0); 16 MFMAs; s_barrier };
//   * the two wave rows run STAGGERED by one barrier (waves 4-7 pass one extra s_barrier up front, waves 0-3 one at the end): in
//     every barrier interval one wave of each SIMD issues MFMAs while its partner issues LDS reads and DMA;
//   * DMA stream (one half-tile per phase), K-tile t = phases 4t+1 .. 4t+4:  phase 4t+1: A1(t+1), 4t+2: B0(t+2), 4t+3: A0(t+2), 4t+4: B1(t+2)
//     then s_waitcnt vmcnt(6): everything but those last three half-tiles has landed = K-tile t+1 is complete; it is first read in phase
//     4t+5, one phase AFTER the wait (RAW rule of the template: the wait precedes the phase's first barrier, every reader passes a later one);
//     WAR: a half-tile is re-filled >= 2 phases after the phase that read it (A0: read 4t+1, filled 4t+3; B1: 4t+2 / 4t+4; A1: 4t+3 / 4t+5),
//     or 1 phase after when the reads were retired ahead of the reading phase's first barrier (B0: read first in phase 4t+1 and retired by
//     lgkmcnt(8) in front of the barrier, filled in 4t+2).  K-tiles past the end are "filled" through a zero-record descriptor (no memory
//     access, zeros land): the vmcnt arithmetic is the same in every iteration.
//   * PERSISTENT tile walk of the NT kernel (round 6): one workgroup per CU walks its XCD's work items; K-tile 0 of the NEXT tile is
//     fetched into LDS buffer 0 under the epilogue of the current one, which stages through buffer 1 only (8 KB per wave: two rounds of
//     64 rows, four of 32 for the fp32-staged gelu' form); behind the epilogue ONE s_waitcnt vmcnt(0) + barrier (loads and stores do not
//     retire in order with respect to each other, so no counted wait there), then the lead of K-tile 1 and the unchanged steady state.
//     The GELU tables are fetched once per workgroup instead of once per tile.  Same box, interleaved (profiles/r06_gemm_persist_ab.txt):
//     dgrad + gelu' 3.63 vs 3.76 ms (-3.6 %), FFN-in + GELU 3.74 vs 3.79 (-1.3 %), plain NT at K = 768 -2.3 ... -3.7 %, at K >= 2304 +-0.4 %.
#include <stdlib.h>

#include "common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef bf16_t gfrag __attribute__((ext_vector_type(8)));

constexpr int G_REGION = 16384;            // one half-tile
constexpr int G_BUF = 4 * G_REGION;        // A0 | A1 | B0 | B1 of one K-tile
constexpr int G_SMEM = 2 * G_BUF;          // 128 KB
constexpr int G_XCDS = 8;
// The GELU epilogues look gelu / gelu' of a bf16 value up in LDS: a COMPACT copy of the 65 536-entry tables covering the bit patterns with
// 2^-24 <= |x| < 16 (28 binades x 128 mantissas x 2 signs = 7 168 entries: 28 KB of fp32 or 14 KB of bf16 behind the 128 KB of tile
// buffers), fetched by LDS-DMA at kernel entry.  Outside it the functions are trivial in fp32: |x| < 2^-24: gelu = x / 2 exactly,
// gelu' = gelu'(2^-24) to 5e-8; |x| >= 16: gelu = x or -0, gelu' = 1 or 0 exactly (the table's edge entries).  A gather from the full table
// in global memory costs the tile ~26 us (64 distinct lines per wave-instruction); erff per element is ~60 VALU instructions.
constexpr int G_TAB_LO = 103 * 128, G_TAB_HI = 131 * 128, G_TAB_RANGE = G_TAB_HI - G_TAB_LO;      // bf16 magnitudes [2^-24, 16)
constexpr int G_SMEM_GRAD = G_SMEM + 2 * G_TAB_RANGE * 4, G_SMEM_DUAL = G_SMEM + 2 * G_TAB_RANGE * 2;
// The bias of a tile's 256 columns goes through LDS as well (round 6): 2 tiles (current / next) x 4 wave columns x 256 B behind the tables,
// one 4-byte-per-lane LDS-DMA per wave and tile, issued with the NEXT tile's K-tile 0.  As 16 conditional global loads per lane at the
// head of the epilogue (the compiled form of `bias ? bias[n + j] : 0`) every one of them stood behind its own s_waitcnt vmcnt(0) - which
// also waited for the K-tile 0 just requested: ~4 us of a 25 us tile at K = 768 (profiles/r06_gemm_bias_lds.txt).
constexpr int G_BIAS_BYTES = 2048;

enum { EPI_STORE = 0, EPI_ACCUM = 1, EPI_GELU_GRAD = 2, EPI_GELU_DUAL = 3, EPI_SLAB = 4 };

struct GemmArgs {
  const bf16_t* A;        // NT: [M, K] row stride lda.            TN: dY [rows, No], row stride lda
  const bf16_t* W;        // NT: [N, K] row stride ldw.            TN: X  [rows, Ni], row stride ldw
  const bf16_t* bias;     // [N] or null (NT)
  bf16_t* C;              // NT: [M, N] row stride ldc
  const bf16_t* U;        // EPI_GELU_GRAD: the FFN pre-activation [M, N] (row stride ldc)
  bf16_t* G;              // EPI_GELU_DUAL: gelu(u) [M, N] (row stride ldc)
  const void* table;      // EPI_GELU_GRAD: gelu'(x) fp32 per bf16 bit pattern; EPI_GELU_DUAL: gelu(x) bf16 per bit pattern
  float* part;            // EPI_GELU_GRAD: column sums [2 * tiles_m][N];  TN: slabs [splits][No][Ni]
  int M, N, K;            // TN: M = No, N = Ni, K = rows of this launch
  int64_t lda, ldw, ldc;
  int tiles_m, tiles_n;
  int ngroup;             // NT: N-tiles per group of the work map (nt_item)
  int k_chunk;            // TN: rows per split (a multiple of 128)
  int splits;             // TN: row splits
};

__device__ __forceinline__ u32x4 g_rsrc(const void* base, uint32_t bytes) {       // buffer descriptor in SCALAR registers
  const uint64_t a = (uint64_t)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) & 0xffffu;
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ uint32_t g_lds_addr(const char* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
// LDS-DMA of one 1 KB piece: M0 = LDS byte address (wave-uniform), lane l lands at M0 + 16 l.  Inline asm: the compiler neither sees a
// VMEM instruction to put its own vmcnt(0) behind, nor re-orders it (volatile + memory clobber); the waits are counted by hand.
__device__ __forceinline__ void g_dma16(u32x4 rs, uint32_t lds, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void g_dma4(u32x4 rs, uint32_t lds, int voff, int soff) {       // 256 B: lane l lands at M0 + 4 l
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
#define G_BAR() asm volatile("s_barrier" ::: "memory")
#define G_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int g_f(int row) { return (row >> 1) & 7; }      // NT image: 16-byte chunk c of row r sits at c ^ f(r)

// ---------------------------------------------------------------------------------------------------------------------------------
// operand loaders.  Both present the same interface to the main loop:
//   stage(region, kt, buf)   DMA of half-tile `region` (0 A0, 1 A1, 2 B0, 3 B1) of K-tile kt (zeros past the end) - 2 pieces per wave
//   ldm(dst, buf, mh) / ldn(dst, buf, nh)   fragment reads of the M-side half mh (4 tiles x 2 k-steps) / N-side half nh (2 x 2)
// ---------------------------------------------------------------------------------------------------------------------------------
struct LoaderNT {
  // A half-tile image: 128 rows x 128 B; row rho of A_h = tile row (rho >> 6) * 128 + h * 64 + (rho & 63); of B_h = tile column
  // (rho >> 5) * 64 + h * 32 + (rho & 31).  Piece j = rows 8 j .. 8 j + 7; wave w stages pieces w and w + 8.
  u32x4 rsa, rsw;
  int va[2][2], vw[2][2];       // [h][piece]: per-lane source offsets (row clamped to the matrix: no reliance on out-of-range semantics)
  uint32_t lds0;                // LDS byte address of smem + wave * 1024
  int KT;
  const char* smem;
  int xo[2];                    // per-lane fragment offsets (k-step 0 / 1)
  int xbase, wbase;             // wave parts: wr * 64 * 128, 2 * G_REGION + wc * 32 * 128

  __device__ __forceinline__ void init(const GemmArgs& p, const char* smem_, int m0, int n0, int wave, int lane) {
    smem = smem_;
    KT = p.K / 64;
    const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, lq = lane >> 4;
    const int r8 = lane >> 3;
    const int rho = wave * 8 + r8;                       // 0 .. 63 (piece w); piece w + 8 is rho + 64
    const int cb = (((lane & 7) ^ g_f(rho)) << 4);       // byte offset of the LOGICAL chunk this lane's LDS slot holds
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        int ra = m0 + pc * 128 + h * 64 + rho;           // A_h: wr' = pc
        ra = ra < p.M ? ra : p.M - 1;
        va[h][pc] = (ra - m0) * (int)p.lda * 2 + cb;
        const int rb = wave * 8 + r8 + pc * 64;          // B_h row 0 .. 127: wc' = rb >> 5
        int cn = n0 + (rb >> 5) * 64 + h * 32 + (rb & 31);
        cn = cn < p.N ? cn : p.N - 1;
        vw[h][pc] = (cn - n0) * (int)p.ldw * 2 + cb;
      }
    rsa = g_rsrc(p.A + (int64_t)m0 * p.lda, 0xffffffffu);
    rsw = g_rsrc(p.W + (int64_t)n0 * p.ldw, 0xffffffffu);
    lds0 = __builtin_amdgcn_readfirstlane(g_lds_addr(smem_) + (uint32_t)wave * 1024u);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xo[ks] = l15 * 128 + (((4 * ks + lq) ^ g_f(l15)) << 4);
    xbase = wr * 64 * 128;
    wbase = 2 * G_REGION + wc * 32 * 128;
  }
  template <int REGION>
  __device__ __forceinline__ void stage(int kt, int buf) {
    constexpr int h = REGION & 1;
    u32x4 rs = REGION < 2 ? rsa : rsw;
    rs[2] = __builtin_amdgcn_readfirstlane(kt < KT ? 0xffffffffu : 0u);      // past the end: zero records, nothing is read (pinned to a scalar register)
    const uint32_t dst = lds0 + (uint32_t)(buf * G_BUF + REGION * G_REGION);
    const int so = kt * 128;
    if constexpr (REGION < 2) {
      g_dma16(rs, dst, va[h][0], so);
      g_dma16(rs, dst + 8192, va[h][1], so);
    } else {
      g_dma16(rs, dst, vw[h][0], so);
      g_dma16(rs, dst + 8192, vw[h][1], so);
    }
  }
  __device__ __forceinline__ void ldm(gfrag (&d)[4][2], int buf, int mh) const {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) d[mt][ks] = *reinterpret_cast<const gfrag*>(smem + buf * G_BUF + mh * G_REGION + xbase + mt * 2048 + xo[ks]);
  }
  __device__ __forceinline__ void ldn(gfrag (&d)[2][2], int buf, int nh) const {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) d[nt][ks] = *reinterpret_cast<const gfrag*>(smem + buf * G_BUF + nh * G_REGION + wbase + nt * 2048 + xo[ks]);
  }
};

struct LoaderTN {
  // A half-tile image: 64 k-rows x 128 columns in 8-row x 32-column subtiles of 512 B (cdna_hip_programming.md T10, image (a)):
  //   off(row, ch) = 2048 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3)),   ch = 8-column chunk 0 .. 15
  // column gamma of A_h = tile-M column (gamma >> 6) * 128 + h * 64 + (gamma & 63); of B_h = tile-N column (gamma >> 5) * 64 + h * 32 + (gamma & 31).
  // Piece p = row block p >> 1, chunk blocks 2 (p & 1), 2 (p & 1) + 1 (8 rows x 128 B of source); wave w stages pieces w and w + 8
  // (k-rows + 32).  The k-row part of the source offset lives in the VECTOR offset, so rows behind the end of the operand fall outside
  // the descriptor's records and read as zeros whatever the hardware does with the scalar offset.
  u32x4 rsa, rsw;
  int va, vw;                   // per-lane source offset of (row 8 (w >> 1) + ..., column of region h = 0) inside the K-chunk
  int stepa, stepw;             // bytes per 32 k-rows
  uint32_t lds0;
  int KT;
  const char* smem;
  int tr[2][2];                 // [parity of the 16-column block][e]: per-lane transposed-read offsets
  int xbase, wbase;

  __device__ __forceinline__ void init(const GemmArgs& p, const char* smem_, int m0, int n0, int wave, int lane, int k0, int krows) {
    smem = smem_;
    KT = (krows + 63) / 64;
    const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, lq = lane >> 4;
    const int sub = lane >> 5, r7 = (lane >> 2) & 7, pos = lane & 3;
    const int row = 8 * (wave >> 1) + r7;                                  // k-row of piece w inside the K-tile (piece w + 8: + 32)
    const int x = pos ^ (((row >> 2) & 3));                                // logical chunk (ch & 3) held by this lane's LDS slot
    const int col_a = (wave & 1) * 128 + 32 * sub + 8 * x;                 // + h * 64
    const int col_b = (2 * (wave & 1) + sub) * 64 + 8 * x;                 // + h * 32
    va = row * (int)p.lda * 2 + col_a * 2;
    vw = row * (int)p.ldw * 2 + col_b * 2;
    stepa = 32 * (int)p.lda * 2;
    stepw = 32 * (int)p.ldw * 2;
    // records: the krows valid rows of this K-chunk (the last row's bytes up to the tile's last column are inside: m0 + 256 <= No <= lda)
    rsa = g_rsrc(p.A + (int64_t)k0 * p.lda + m0, (uint32_t)((int64_t)krows * p.lda * 2 - (int64_t)m0 * 2));
    rsw = g_rsrc(p.W + (int64_t)k0 * p.ldw + n0, (uint32_t)((int64_t)krows * p.ldw * 2 - (int64_t)n0 * 2));
    lds0 = __builtin_amdgcn_readfirstlane(g_lds_addr(smem_) + (uint32_t)wave * 1024u);
    const int q = l15 >> 2, pp = l15 & 3, pb = pp >> 1;
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int e = 0; e < 2; ++e) tr[par][e] = 2048 * lq + 64 * q + 8 * (pp & 1) + 32 * (par ^ (lq & 1)) + 16 * (pb ^ e) + 256 * e;
    xbase = 1024 * wr;                       // 512 * (2 wr + (mt >> 1))
    wbase = 2 * G_REGION + 512 * wc;
  }
  template <int REGION>
  __device__ __forceinline__ void stage(int kt, int buf) {
    constexpr int h = REGION & 1;
    u32x4 rs = REGION < 2 ? rsa : rsw;
    rs[2] = __builtin_amdgcn_readfirstlane(kt < KT ? rs[2] : 0u);
    const uint32_t dst = lds0 + (uint32_t)(buf * G_BUF + REGION * G_REGION);
    if constexpr (REGION < 2) {
      const int v = va + kt * 2 * stepa + h * 128;
      g_dma16(rs, dst, v, 0);
      g_dma16(rs, dst + 8192, v + stepa, 0);
    } else {
      const int v = vw + kt * 2 * stepw + h * 64;
      g_dma16(rs, dst, v, 0);
      g_dma16(rs, dst + 8192, v + stepw, 0);
    }
  }
  __device__ __forceinline__ gfrag trd(const char* base, int par) const {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + tr[par][0]));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + tr[par][1]));
    const s16x8 c = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(gfrag, c);
  }
  __device__ __forceinline__ void ldm(gfrag (&d)[4][2], int buf, int mh) const {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) d[mt][ks] = trd(smem + buf * G_BUF + mh * G_REGION + xbase + 512 * (mt >> 1) + 8192 * ks, mt & 1);
  }
  __device__ __forceinline__ void ldn(gfrag (&d)[2][2], int buf, int nh) const {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) d[nt][ks] = trd(smem + buf * G_BUF + nh * G_REGION + wbase + 8192 * ks, nt);
  }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// the main loop: acc[mt][nt] (C^T tiles: register j of a lane = output column nt * 16 + 4 (lane >> 4) + j of output row mt * 16 + (lane & 15))
// += over K-tiles 0 .. KT-1 (KT even).  On return every DMA has landed, every wave has passed a common barrier: LDS is free.
// ---------------------------------------------------------------------------------------------------------------------------------
template <class Loader>
__device__ __forceinline__ void gemm_mainloop(Loader& ld, f32x4 (&acc)[8][4], int KT, int wave_row, bool prefetched = false) {
  gfrag xf[4][2], wa[2][2], wb[2][2];
#define G_MMA(xf_, wf_, mh_, nh_)                                                                                                  \
  __builtin_amdgcn_s_setprio(1);                                                                                                   \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) \
    acc[(mh_) * 4 + mt][(nh_) * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf_[nt][ks], xf_[mt][ks], acc[(mh_) * 4 + mt][(nh_) * 2 + nt], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);
  // prologue: K-tile 0 whole, then B0 A0 B1 of K-tile 1 (the steady-state lead); K-tile 0 has landed behind vmcnt(6).
  // PERSISTENT tile walk (round 6): `prefetched` = K-tile 0 of this tile was issued under the previous tile's epilogue and has landed
  // (the caller's vmcnt(0) + barrier): only the lead of K-tile 1 is issued here, and nothing is waited for
  if (!prefetched) {
    ld.template stage<0>(0, 0);
    ld.template stage<2>(0, 0);
    ld.template stage<3>(0, 0);
    ld.template stage<1>(0, 0);
  }
  ld.template stage<2>(1, 1);
  ld.template stage<0>(1, 1);
  ld.template stage<3>(1, 1);
  if (!prefetched) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  G_BAR();
  if (wave_row == 1) G_BAR();                         // the stagger: waves 4-7 run one barrier behind
  for (int kt = 0; kt < KT; kt += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int t = kt + half;                        // K-tile of these four phases, in buffer `half` (KT even)
      // ---- phase 1: quadrant (m0, n0); reads B0 (retired ahead of the barrier: B0 is re-filled next phase) then A0; DMA A1(t + 1)
      ld.ldn(wa, half, 0);
      G_FENCE();
      ld.ldm(xf, half, 0);
      G_FENCE();
      ld.template stage<1>(t + 1, half ^ 1);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      G_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      G_FENCE();
      G_MMA(xf, wa, 0, 0);
      G_FENCE();
      G_BAR();
      // ---- phase 2: quadrant (m0, n1); reads B1; DMA B0(t + 2)
      ld.ldn(wb, half, 1);
      G_FENCE();
      ld.template stage<2>(t + 2, half);
      G_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      G_FENCE();
      G_MMA(xf, wb, 0, 1);
      G_FENCE();
      G_BAR();
      // ---- phase 3: quadrant (m1, n1); reads A1 (over the A0 fragments, which phase 2's MFMAs have consumed); DMA A0(t + 2)
      ld.ldm(xf, half, 1);
      G_FENCE();
      ld.template stage<0>(t + 2, half);
      G_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      G_FENCE();
      G_MMA(xf, wb, 1, 1);
      G_FENCE();
      G_BAR();
      // ---- phase 4: quadrant (m1, n0), operands in registers; DMA B1(t + 2); K-tile t + 1 has landed behind vmcnt(6)
      ld.template stage<3>(t + 2, half);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      G_BAR();
      G_FENCE();
      G_MMA(xf, wa, 1, 0);
      G_FENCE();
      if (t + 1 == KT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing zero fills, ahead of this wave's last barrier
      G_BAR();
    }
  }
  if (wave_row == 0) G_BAR();
#undef G_MMA
}

// ---------------------------------------------------------------------------------------------------------------------------------
// NT kernel
// ---------------------------------------------------------------------------------------------------------------------------------
// Work map of the NT kernel.  XCD x (workgroups with id % 8 == x, dispatched in id order, 32 at a time) owns a contiguous range of
// M-blocks.  Inside it the N-tiles are taken in GROUPS of b: group-major, then M-block, then the N-tile inside the group - so the 32
// workgroups an XCD runs at a time cover 32 / b M-blocks x b N-tiles, the b weight panels of the group (b x 384 KB at K = 768) stay in
// that XCD's L2 for the whole pass over the M range, and an activation panel is shared by b workgroups that start together.
// Measured on the step's shapes (profiles/r05_gemm_nt_map.txt, M = 649 984, N = 3072, K = 768; FETCH_SIZE per launch | time):
//   b = 12 (all N-tiles of an M-block side by side: the 12 weight panels, 4.7 MB, do not fit the 4 MB L2 and are fetched again from the
//   Infinity Cache by every round of workgroups)  5.9 GB | 3.92 ms with the dual epilogue, 16.6 GB | 3.89 ms with the gelu' epilogue;
//   b = 6  4.8 GB | 3.96 ms, 10.9 GB | 3.85 ms;   b = 4  4.5 GB | 4.03 ms, 8.7 GB | 3.97 ms;   b = 1  12.4 GB | 4.61 ms.
// Fewer bytes, not less time: these launches are bound by the matrix pipe at its power-limited clock and by their epilogues, not by
// operand fetch - so the default stays b = tiles_n (T2S_GEMM_NT_GROUP overrides it for probe runs).
__device__ __forceinline__ bool nt_item(int wg, int tiles_m, int tiles_n, int b, int& tm, int& tn) {
  const int x = wg % G_XCDS, j = wg / G_XCDS;
  const int mbx = (tiles_m + G_XCDS - 1) / G_XCDS;
  const int m_lo = x * mbx;
  const int mcount = tiles_m - m_lo < mbx ? tiles_m - m_lo : mbx;
  if (mcount <= 0) return false;
  const int full = tiles_n / b, rem = tiles_n - full * b;
  const int in_full = full * mcount * b;
  if (j < in_full) {
    const int per = mcount * b, g = j / per, r = j - g * per;
    tm = m_lo + r / b;
    tn = g * b + (r - (r / b) * b);
    return true;
  }
  const int jr = j - in_full;
  if (rem == 0 || jr >= mcount * rem) return false;
  tm = m_lo + jr / rem;
  tn = full * b + (jr - (jr / rem) * rem);
  return true;
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, lq = lane >> 4;
  // PERSISTENT tile walk (round 6): one workgroup per CU; the workgroups of XCD x (blockIdx % 8) take the items of that XCD's
  // range in turn (item j = slot, slot + per_xcd, ...: at any time the XCD's CUs hold consecutive items, as the one-tile-per-workgroup
  // dispatch did), and K-tile 0 of the NEXT tile is fetched into LDS buffer 0 while the epilogue of the current one drains through
  // buffer 1 - the 7-half-tile prologue no longer stands alone in front of every 12-K-tile main loop.
  const int xcd = (int)(blockIdx.x % G_XCDS), per_xcd = (int)(gridDim.x / G_XCDS);
  int item = (int)(blockIdx.x / G_XCDS);
  int tm, tn;
  if (!nt_item(item * G_XCDS + xcd, p.tiles_m, p.tiles_n, p.ngroup, tm, tn)) return;      // workgroup-uniform
  tm = __builtin_amdgcn_readfirstlane(tm);
  tn = __builtin_amdgcn_readfirstlane(tn);

  if constexpr (EPI == EPI_GELU_GRAD || EPI == EPI_GELU_DUAL) {
    // the compact table: [sign][G_TAB_RANGE] entries behind the tile buffers; 1 KB pieces round the waves, ONCE per workgroup.  These
    // DMAs are OLDER than every tile DMA, so the first main loop's counted waits cover them, and its barriers publish them
    constexpr int ES = EPI == EPI_GELU_GRAD ? 4 : 2, HALF = G_TAB_RANGE * ES, PIECES = 2 * HALF / 1024;
    const u32x4 rs = g_rsrc(p.table, 65536u * ES);
    const uint32_t dst = __builtin_amdgcn_readfirstlane(g_lds_addr(smem + G_SMEM));
    for (int pc = wave; pc < PIECES; pc += 8) {
      const int half = pc >= PIECES / 2 ? 1 : 0;
      g_dma16(rs, dst + (uint32_t)pc * 1024u, lane * 16, (half * 0x8000 + G_TAB_LO) * ES + (pc - half * (PIECES / 2)) * 1024);
    }
  }
  // bias of the tile's columns: wave column wc's 64 values (+ the 64 behind them, unused) into slot [tile parity][wc]; columns >= N and a
  // null bias read as zeros (descriptor range).  The first tile's DMA is older than every tile DMA: the main loop's waits and barriers cover it
  constexpr bool HAS_BIAS = EPI != EPI_GELU_GRAD;
  constexpr int BIAS_OFF = EPI == EPI_GELU_DUAL ? G_SMEM_DUAL : G_SMEM;
  const u32x4 rs_bias = g_rsrc(p.bias, p.bias ? (uint32_t)p.N * 2u : 0u);
  const uint32_t bias_lds = __builtin_amdgcn_readfirstlane(g_lds_addr(smem + BIAS_OFF) + (uint32_t)wc * 256u);
  int bias_buf = 0;
  if constexpr (HAS_BIAS) g_dma4(rs_bias, bias_lds, (tn * 256 + wc * 64) * 2 + lane * 4, 0);     // (the VECTOR offset is the range-checked one)
  LoaderNT ld;
  ld.init(p, smem, tm * 256, tn * 256, wave, lane);
  bool prefetched = false;
  for (;;) {
  const int m0 = tm * 256, n0 = tn * 256;
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_mainloop(ld, acc, p.K / 64, wr, prefetched);
  // the next item of this workgroup; its K-tile 0 goes out NOW, into buffer 0 (on return from the main loop every DMA has landed and
  // every wave has passed a common barrier: LDS is free), under the epilogue below, which stages through buffer 1 only
  item += per_xcd;
  int tm2 = 0, tn2 = 0;
  const bool more = nt_item(item * G_XCDS + xcd, p.tiles_m, p.tiles_n, p.ngroup, tm2, tn2);       // workgroup-uniform
  if (more) {
    tm = __builtin_amdgcn_readfirstlane(tm2);
    tn = __builtin_amdgcn_readfirstlane(tn2);
    ld.init(p, smem, tm * 256, tn * 256, wave, lane);
    ld.template stage<0>(0, 0);
    ld.template stage<2>(0, 0);
    ld.template stage<3>(0, 0);
    ld.template stage<1>(0, 0);
    if constexpr (HAS_BIAS) g_dma4(rs_bias, bias_lds + (uint32_t)(bias_buf ^ 1) * 1024u, (tn * 256 + wc * 64) * 2 + lane * 4, 0);
  }

  // ---- epilogue.  The wave's 128 x 64 outputs go through its own 8 KB of LDS in BUFFER 1 (buffer 0 is receiving the next tile's K-tile 0):
  // bf16 forms: two rounds of 64 rows x 128 B (16-byte chunk c of row r at c ^ (r & 7)); the gelu' form stages fp32: four rounds of 32
  // rows x 256 B.  Whole 128-byte row segments leave: 8 rows per wave-instruction.
  char* const cw = smem + G_BUF + wave * 8192;
  const int mw = m0 + wr * 128, nw = n0 + wc * 64;                     // first row / column of this wave's outputs
  const int er = lane >> 3, ec = lane & 7;                             // read-back: row er of every 8-row group, chunk ec (columns 8 ec .. 8 ec + 7)
  if constexpr (EPI == EPI_STORE || EPI == EPI_ACCUM || EPI == EPI_GELU_DUAL) {
    float b4[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const bf16x4 bb = *reinterpret_cast<const bf16x4*>(smem + BIAS_OFF + bias_buf * 1024 + wc * 256 + (nt * 16 + 4 * lq) * 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) b4[nt][j] = (float)bb[j];
    }
    const bool cols_ok = nw + ec * 8 + 7 < p.N;
    const unsigned short* ltab = reinterpret_cast<const unsigned short*>(smem + G_SMEM);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int row = mt * 16 + l15;
        const f32x4 v = acc[round * 4 + mt][nt];
        const bf16x4 o = {(bf16_t)(v[0] + b4[nt][0]), (bf16_t)(v[1] + b4[nt][1]), (bf16_t)(v[2] + b4[nt][2]), (bf16_t)(v[3] + b4[nt][3])};
        *reinterpret_cast<bf16x4*>(cw + row * 128 + (((nt * 2 + (lq >> 1)) ^ (row & 7)) << 4) + (lq & 1) * 8) = o;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // (own region only: no workgroup barrier)
#pragma unroll 4
    for (int i = 0; i < 8; ++i) {
      const int row = i * 8 + er;
      typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
      u16x8 v = *reinterpret_cast<const u16x8*>(cw + row * 128 + ((ec ^ (row & 7)) << 4));
      const int m = mw + round * 64 + row;
      if (m < p.M && cols_ok) {
        bf16_t* dst = p.C + (int64_t)m * p.ldc + nw + ec * 8;
        if constexpr (EPI == EPI_ACCUM) {
          const bf16x8 old = *reinterpret_cast<const bf16x8*>(dst);
          const bf16x8 add = __builtin_bit_cast(bf16x8, v);
          bf16x8 s;
#pragma unroll
          for (int j = 0; j < 8; ++j) s[j] = (bf16_t)((float)old[j] + (float)add[j]);
          *reinterpret_cast<bf16x8*>(dst) = s;
        } else {
          *reinterpret_cast<u16x8*>(dst) = v;
        }
        if constexpr (EPI == EPI_GELU_DUAL) {
          // g = gelu(u) of the ROUNDED pre-activation: bit-equal to gelu_fwd_kernel on the same u (table values are that kernel's
          // arithmetic; below 2^-24 it yields x / 2 exactly, from 16 up x * (1 + erff) = x or x * 0).  Table offsets of TWO values per
          // packed 16-bit instruction; values outside the table's range are recognised by the largest unclamped offset of the lane's 8
          // values and redone by the per-value arithmetic behind a branch the wave skips.
          typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
          // (the pairs are built from the ELEMENTS of v: __builtin_bit_cast of one element of a re-typed vector is narrowed by this clang to
          // a load of element 0 that stands in for all four - seen in the ISA, as in attn_bwd_fused_bf16.hip)
          const char* const tb = reinterpret_cast<const char*>(ltab);
          u16x8 g;
          u16x2 worst = {0, 0};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const u16x2 b2 = {v[2 * k], v[2 * k + 1]};
            const u16x2 d2u = (b2 & (unsigned short)0x7fff) - (unsigned short)G_TAB_LO;          // wraps below the table, >= RANGE above it
            worst = __builtin_elementwise_max(worst, d2u);
            const u16x2 d2c = __builtin_elementwise_min(d2u, (u16x2)(unsigned short)(G_TAB_RANGE - 1));
            const u16x2 off2 = ((b2 >> (unsigned short)15) * (unsigned short)G_TAB_RANGE + d2c) << (unsigned short)1;      // byte offsets: < 2 * 7168
            const uint32_t ow = __builtin_bit_cast(uint32_t, off2);
            g[2 * k] = *reinterpret_cast<const unsigned short*>(tb + (ow & 0xffffu));
            g[2 * k + 1] = *reinterpret_cast<const unsigned short*>(tb + (ow >> 16));
          }
          if (__builtin_expect(worst[0] >= G_TAB_RANGE || worst[1] >= G_TAB_RANGE, 0)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const uint32_t b = v[j], mag = b & 0x7fffu, sgn = b >> 15;
              const uint32_t d = mag - (uint32_t)G_TAB_LO;
              const float x = __builtin_bit_cast(float, b << 16);
              const float o = mag < (uint32_t)G_TAB_LO ? 0.5f * x : x * (sgn ? 0.f : 1.f);
              if (d >= (uint32_t)G_TAB_RANGE) g[j] = __builtin_bit_cast(unsigned short, (bf16_t)o);
            }
          }
          *reinterpret_cast<u16x8*>(p.G + (int64_t)m * p.ldc + nw + ec * 8) = g;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the round's read-back is in registers before the next round's writes
    }
  } else if constexpr (EPI == EPI_GELU_GRAD) {
    // du = acc * gelu'(u) in fp32 (one rounding), + column sums of du (the FFN bias gradient) per 128-row group.  fp32 staging: rows of
    // 256 B, 16-byte chunk c at c ^ (r & 15); read-back: lane = row (lane >> 3) of every 8-row group, columns 8 (lane & 7) .. + 7: 16-byte
    // loads of u and stores of du, whole 128-byte row segments.  The round's u rows are requested BEFORE the staging pass.
    const float* ltab = reinterpret_cast<const float*>(smem + G_SMEM);
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    float csum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[j] = 0.f;
    const bool cols_ok = nw + ec * 8 + 7 < p.N;
#pragma unroll
    for (int round = 0; round < 4; ++round) {
      u16x8 uv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mw + round * 32 + i * 8 + er;
        uv[i] = (m < p.M && cols_ok) ? *reinterpret_cast<const u16x8*>(p.U + (int64_t)m * p.ldc + nw + ec * 8) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int row = mt * 16 + l15;
          *reinterpret_cast<f32x4*>(cw + row * 256 + (((nt * 4 + lq) ^ (row & 15)) << 4)) = acc[round * 2 + mt][nt];
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + er;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(cw + row * 256 + (((2 * ec) ^ (row & 15)) << 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(cw + row * 256 + (((2 * ec + 1) ^ (row & 15)) << 4));
        const int m = mw + round * 32 + row;
        float d[8];
        {
          // table offsets of two values per packed 16-bit instruction (saturating subtract, min, shift, mad); the derivative is constant
          // outside the table's range (0.5 below 2^-24, 0 or 1 from 16 up), so the clamped offset is exact everywhere
          // A non-finite pre-activation (magnitude bits >= 0x7f80) must poison its gradient as the standalone gelu_bwd pass does
          // (erff / expf of NaN; inf * exp(-inf)): the lane tracks its largest magnitude (one packed max per pair) and redoes such
          // values behind a branch the wave skips - a diverged forward stays visible in du and in the bias-gradient column sums.
          typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
          const char* const tb = reinterpret_cast<const char*>(ltab);
          u16x2 worst = {0, 0};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const u16x2 b2 = {uv[i][2 * k], uv[i][2 * k + 1]};      // (from the elements: see the dual epilogue)
            const u16x2 mag2 = b2 & (unsigned short)0x7fff;
            worst = __builtin_elementwise_max(worst, mag2);
            const u16x2 d2 = __builtin_elementwise_min(__builtin_elementwise_sub_sat(mag2, (u16x2)(unsigned short)G_TAB_LO),
                                                       (u16x2)(unsigned short)(G_TAB_RANGE - 1));
            const u16x2 off2 = ((b2 >> (unsigned short)15) * (unsigned short)G_TAB_RANGE + d2) << (unsigned short)2;       // byte offsets: < 4 * 7168
            const uint32_t ow = __builtin_bit_cast(uint32_t, off2);
            const float t0 = *reinterpret_cast<const float*>(tb + (ow & 0xffffu));
            const float t1 = *reinterpret_cast<const float*>(tb + (ow >> 16));
            d[2 * k] = (k < 2 ? v0[(2 * k) & 3] : v1[(2 * k) & 3]) * t0;
            d[2 * k + 1] = (k < 2 ? v0[(2 * k + 1) & 3] : v1[(2 * k + 1) & 3]) * t1;
          }
          if (__builtin_expect(worst[0] >= 0x7f80 || worst[1] >= 0x7f80, 0)) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if ((uv[i][j] & 0x7fffu) >= 0x7f80u) d[j] = __builtin_nanf("");
          }
        }
        if (m < p.M && cols_ok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) csum[j] += d[j];
          const bf16x8 o = {(bf16_t)d[0], (bf16_t)d[1], (bf16_t)d[2], (bf16_t)d[3], (bf16_t)d[4], (bf16_t)d[5], (bf16_t)d[6], (bf16_t)d[7]};
          *reinterpret_cast<bf16x8*>(p.C + (int64_t)m * p.ldc + nw + ec * 8) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // column sums of this wave's 128 rows: the lanes with equal (lane & 7) hold the same 8 columns
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      csum[j] += __shfl_xor(csum[j], 8, 64);
      csum[j] += __shfl_xor(csum[j], 16, 64);
      csum[j] += __shfl_xor(csum[j], 32, 64);
    }
    if (lane < 8 && cols_ok) {
      float* dst = p.part + (int64_t)((m0 >> 8) * 2 + wr) * p.N + nw + ec * 8;
      *reinterpret_cast<f32x4*>(dst) = f32x4{csum[0], csum[1], csum[2], csum[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{csum[4], csum[5], csum[6], csum[7]};
    }
  }
  if (!more) break;
  // K-tile 0 of the next tile and its bias have landed (this wave's pieces; the epilogue's global stores are behind the same wait) and every
  // wave is done with its staging area: the main loop may run again, buffer 1 may be filled again.  (Vector-memory operations retire in issue
  // order and the next tile's DMAs are OLDER than the epilogue's accesses, so a counted wait - vmcnt(16 / 32 / 34) on interior tiles - would
  // leave the stores in flight under the next K-tile 0: measured, same box, no difference at any shape - 2.861 vs 2.869 ms at N = 3072,
  // K = 768 (profiles/r06_gemm_bias_lds.txt) - the store drain is not what a tile waits for; the plain wait stays.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  prefetched = true;
  bias_buf ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// TN kernel (weight gradient): workgroup = (tile, row split); fp32 slab [split][No][Ni]
// ---------------------------------------------------------------------------------------------------------------------------------
// Which (row split, tile) a workgroup takes.  The 256 x 256 tile moves 64 KB per K-tile for 8.4 MFLOP: ~40 GB/s per CU at 1.25 PFLOP/s,
// more than a CU draws from the Infinity Cache (~33 GB/s, MI355X_MICROARCH.md "Indexed rows") - the operands must come from the XCD's
// L2.  With (split, tile) = (id / tiles, id % tiles) the tiles of one row split - which read the SAME rows - were dealt round-robin to
// all 8 XCDs (id % 8) and the kernel ran at the Infinity-Cache rate (24 GB of L2 fills in 2.7 ms, profiles/r05_gemm_probe_v1.txt).
// Here the workgroups of one XCD (ids equal mod 8, c = grid / 8 of them, one per CU) take tiles of ONE row split as far as that goes:
//   T <= c:  every XCD hosts c / T whole splits; the splits left over are dealt to the spare slots in id order
//   T >  c:  XCD x < S hosts c tiles of split x; the T - c tiles left of every split go to the XCDs S .. 7
// Placement is a speed matter only: every (split, tile) is taken exactly once whatever the hardware does (tests/test_gemm_cpu.py
// enumerates this function's Python mirror).  Returns false for a workgroup without an item.
__device__ __forceinline__ bool tn_item(int wg, int c, int T, int S, int& split, int& tile) {
  const int x = wg % G_XCDS, j = wg / G_XCDS;
  if (T <= c) {
    const int spx = c / T, hosted = spx * T;
    if (j < hosted) {
      split = x * spx + j / T;
      tile = j % T;
      return split < S;
    }
    const int q = x * (c - hosted) + (j - hosted);
    split = G_XCDS * spx + q / T;
    tile = q % T;
    return split < S;
  }
  if (x < S) {
    split = x;
    tile = j;
    return j < c;
  }
  const int L = T - c, q = (x - S) * c + j;
  split = q / L;
  tile = c + q % L;
  return split < S;
}

__global__ __launch_bounds__(512, 2) void gemm_tn_bf16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, lq = lane >> 4;
  const int ntiles = p.tiles_m * p.tiles_n;
  int tile, split;
  if (!tn_item((int)blockIdx.x, (int)gridDim.x / G_XCDS, ntiles, p.splits, split, tile)) return;      // workgroup-uniform
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int k0 = split * p.k_chunk;
  int krows = p.K - k0;
  krows = krows < p.k_chunk ? krows : p.k_chunk;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (krows > 0) {
    LoaderTN ld;
    ld.init(p, smem, m0, n0, wave, lane, k0, krows);
    gemm_mainloop(ld, acc, p.k_chunk / 64, wr);
  }
  // slab store: lane owns 4 consecutive Ni columns (registers) of No row l15 of each 16 x 16 tile
  float* const slab = p.part + (int64_t)split * p.M * p.N;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      *reinterpret_cast<f32x4*>(slab + (int64_t)(m0 + wr * 128 + mt * 16 + l15) * p.N + n0 + wc * 64 + nt * 16 + 4 * lq) = acc[mt][nt];
}

// out[i] (= or +=) sum over splits of slab[s][i], in split order: bit-reproducible
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t n4, int splits, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 s = accumulate ? *reinterpret_cast<const f32x4*>(out + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(slabs + (int64_t)k * n4 * 4 + i * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = s;
  }
}

// exact-erf GELU and its derivative, the arithmetic of gelu.hip's standalone kernels, evaluated once per bf16 value
__device__ __forceinline__ float g_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float g_gelu_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__global__ __launch_bounds__(256) void gelu_tables_kernel(bf16_t* __restrict__ fwd, float* __restrict__ grad) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;          // 65 536 threads: one per bf16 bit pattern
  const float x = __builtin_bit_cast(float, i << 16);
  if (fwd) fwd[i] = (bf16_t)g_gelu(x);
  if (grad) grad[i] = g_gelu_grad(x);
}

// > 64 KB of dynamic LDS needs the opt-in, and the attribute is PER DEVICE: one flag per (kernel form, device ordinal) - the pattern of
// launch_attn_bwd_fused_bf16 - so that a process driving a second GPU opts in there too (an ordinal outside the table: set it every call)
struct LdsFlags { bool done[64]; };
template <typename K>
int g_reserve_lds(K kernel, LdsFlags& f, const char* what, int bytes = G_SMEM) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev >= 0 && f.done[dev]) return 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    t2s_set_error("%s: cannot reserve %d bytes of LDS per workgroup", what, bytes);
    return 3;
  }
  if (dev >= 0) f.done[dev] = true;
  return 0;
}

}  // namespace

extern "C" int t2s_gelu_tables(void* fwd_bf16, void* grad_f32, t2s_stream_t stream) {
  T2S_CHECK_ARG(fwd_bf16 || grad_f32, "gelu_tables: null pointers");
  hipLaunchKernelGGL(gelu_tables_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (bf16_t*)fwd_bf16, (float*)grad_f32);
  T2S_CHECK_LAUNCH("gelu_tables");
  return 0;
}

extern "C" int t2s_gemm_nt_colsum_rows(int64_t M) { return (int)(2 * ((M + 255) / 256)); }

extern "C" int t2s_gemm_nt(const void* a, const void* w, const void* bias, void* c, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                           int64_t ldc, int epilogue, const void* u, void* g, const void* table, float* colsum_part, t2s_stream_t stream) {
  T2S_CHECK_ARG(a && w && c, "gemm_nt: null pointer");
  T2S_CHECK_ARG(M > 0 && N > 0 && K > 0 && M < ((int64_t)1 << 31), "gemm_nt: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  T2S_CHECK_ARG(K % 128 == 0, "gemm_nt: K = %d must be a multiple of 128 (two 64-deep K-tiles per loop trip)", K);
  T2S_CHECK_ARG(N % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0, "gemm_nt: N, lda, ldw, ldc must be multiples of 8 (16-byte accesses)");
  T2S_CHECK_ARG(lda >= K && ldw >= K && ldc >= N, "gemm_nt: a row stride is shorter than its row");
  T2S_CHECK_ARG(lda * 512 < ((int64_t)1 << 31) && ldw * 512 < ((int64_t)1 << 31), "gemm_nt: 256 operand rows must span < 2^31 bytes");
  T2S_CHECK_ARG(epilogue >= EPI_STORE && epilogue <= EPI_GELU_DUAL, "gemm_nt: epilogue %d", epilogue);
  T2S_CHECK_ARG(epilogue != EPI_GELU_GRAD || (u && table && colsum_part && !bias), "gemm_nt: the gelu' epilogue needs u, the derivative table and the column-sum buffer (and takes no bias)");
  T2S_CHECK_ARG(epilogue != EPI_GELU_DUAL || (g && table), "gemm_nt: the dual epilogue needs the output g and the gelu table");
  T2S_CHECK_ARG(epilogue != EPI_ACCUM || !bias, "gemm_nt: the accumulate epilogue takes no bias");
  GemmArgs p = {};
  p.A = (const bf16_t*)a; p.W = (const bf16_t*)w; p.bias = (const bf16_t*)bias; p.C = (bf16_t*)c;
  p.U = (const bf16_t*)u; p.G = (bf16_t*)g; p.table = table; p.part = colsum_part;
  p.M = (int)M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
  p.tiles_m = (int)((M + 255) / 256);
  p.tiles_n = (N + 255) / 256;
  // N-tiles per group of the work map (nt_item): all of them (the measured optimum in TIME, see nt_item); T2S_GEMM_NT_GROUP overrides it (probe runs)
  p.ngroup = p.tiles_n;
  if (const char* e = getenv("T2S_GEMM_NT_GROUP")) {
    const int v = atoi(e);
    if (v >= 1) p.ngroup = v < p.tiles_n ? v : p.tiles_n;
  }
  // persistent tile walk: one workgroup per CU (cus / 8 per XCD), each walking its XCD's items; fewer when the XCD's range is shorter
  const int64_t items_x = (int64_t)((p.tiles_m + G_XCDS - 1) / G_XCDS) * p.tiles_n;          // items of the longest XCD range
  T2S_CHECK_ARG(items_x * G_XCDS < ((int64_t)1 << 31), "gemm_nt: too many tiles");
  int64_t per_xcd = 32;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v >= G_XCDS) per_xcd = v / G_XCDS;
  }
  if (const char* e = getenv("T2S_GEMM_NT_PER_XCD")) { const int v = atoi(e); if (v >= 1) per_xcd = v; }
  if (per_xcd > items_x) per_xcd = items_x;
  const int64_t grid = (int64_t)G_XCDS * per_xcd;
  static LdsFlags done[4] = {};
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  switch (epilogue) {
    case EPI_STORE:
      if ((rc = g_reserve_lds(gemm_nt_bf16_kernel<EPI_STORE>, done[0], "gemm_nt", G_SMEM + G_BIAS_BYTES))) return rc;
      hipLaunchKernelGGL(gemm_nt_bf16_kernel<EPI_STORE>, dim3((unsigned)grid), dim3(512), G_SMEM + G_BIAS_BYTES, st, p);
      break;
    case EPI_ACCUM:
      if ((rc = g_reserve_lds(gemm_nt_bf16_kernel<EPI_ACCUM>, done[1], "gemm_nt", G_SMEM + G_BIAS_BYTES))) return rc;
      hipLaunchKernelGGL(gemm_nt_bf16_kernel<EPI_ACCUM>, dim3((unsigned)grid), dim3(512), G_SMEM + G_BIAS_BYTES, st, p);
      break;
    case EPI_GELU_GRAD:
      if ((rc = g_reserve_lds(gemm_nt_bf16_kernel<EPI_GELU_GRAD>, done[2], "gemm_nt", G_SMEM_GRAD))) return rc;
      hipLaunchKernelGGL(gemm_nt_bf16_kernel<EPI_GELU_GRAD>, dim3((unsigned)grid), dim3(512), G_SMEM_GRAD, st, p);
      break;
    default:
      if ((rc = g_reserve_lds(gemm_nt_bf16_kernel<EPI_GELU_DUAL>, done[3], "gemm_nt", G_SMEM_DUAL + G_BIAS_BYTES))) return rc;
      hipLaunchKernelGGL(gemm_nt_bf16_kernel<EPI_GELU_DUAL>, dim3((unsigned)grid), dim3(512), G_SMEM_DUAL + G_BIAS_BYTES, st, p);
      break;
  }
  T2S_CHECK_LAUNCH("gemm_nt");
  return 0;
}

// row splits of the weight-gradient GEMM: one round of workgroups on the card's CUs
extern "C" int t2s_gemm_wgrad_splits(int64_t rows, int n_out, int n_in) {
  if (rows <= 0 || n_out <= 0 || n_in <= 0 || n_out % 256 || n_in % 256) return 0;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int tiles = (n_out / 256) * (n_in / 256);
  int splits = cus / tiles;
  if (splits < 1) splits = 1;
  const int64_t max_splits = (rows + 127) / 128;
  if (splits > max_splits) splits = (int)max_splits;
  return splits;
}

extern "C" int t2s_gemm_wgrad(const void* dy, const void* x, float* dw, float* slabs, int64_t rows, int n_out, int n_in, int64_t ld_dy,
                              int64_t ld_x, int splits, int accumulate, t2s_stream_t stream) {
  T2S_CHECK_ARG(dy && x && dw && slabs, "gemm_wgrad: null pointer");
  T2S_CHECK_ARG(rows > 0 && rows < ((int64_t)1 << 31) && n_out > 0 && n_in > 0, "gemm_wgrad: bad shape");
  T2S_CHECK_ARG(n_out % 256 == 0 && n_in % 256 == 0, "gemm_wgrad: n_out = %d and n_in = %d must be multiples of 256", n_out, n_in);
  T2S_CHECK_ARG(ld_dy >= n_out && ld_x >= n_in && ld_dy % 8 == 0 && ld_x % 8 == 0, "gemm_wgrad: row strides must cover the rows and be multiples of 8");
  T2S_CHECK_ARG(splits >= 1 && splits <= 4096, "gemm_wgrad: splits = %d", splits);
  int64_t chunk = (rows + splits - 1) / splits;
  chunk = (chunk + 127) / 128 * 128;
  T2S_CHECK_ARG(chunk * ld_dy * 2 < ((int64_t)1 << 32) && chunk * ld_x * 2 < ((int64_t)1 << 32), "gemm_wgrad: a row split must span < 4 GB of each operand");
  T2S_CHECK_ARG((chunk + 64) * ld_dy * 2 < ((int64_t)1 << 31) && (chunk + 64) * ld_x * 2 < ((int64_t)1 << 31),
                "gemm_wgrad: a row split must span < 2 GB of each operand (32-bit vector offsets): use more splits");
  GemmArgs p = {};
  p.A = (const bf16_t*)dy; p.W = (const bf16_t*)x; p.part = slabs;
  p.M = n_out; p.N = n_in; p.K = (int)rows; p.lda = ld_dy; p.ldw = ld_x;
  p.tiles_m = n_out / 256; p.tiles_n = n_in / 256;
  p.k_chunk = (int)chunk;
  p.splits = splits;
  // grid: c workgroups per XCD with 8 c >= tiles * splits and, when a split has more tiles than c, room for the left-over tiles
  const int items = p.tiles_m * p.tiles_n * splits;
  int c = (items + G_XCDS - 1) / G_XCDS;
  {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (items <= cus && c < cus / G_XCDS) c = cus / G_XCDS;          // one round on the card: the XCD-shaped map (32 per XCD)
  }
  const int T = p.tiles_m * p.tiles_n;
  T2S_CHECK_ARG(T <= c ? (G_XCDS * (c / T) >= splits || (int64_t)G_XCDS * (c - (c / T) * T) >= (int64_t)(splits - G_XCDS * (c / T)) * T)
                       : (splits <= G_XCDS && (int64_t)(G_XCDS - splits) * c >= (int64_t)splits * (T - c)),
                "gemm_wgrad: %d tiles x %d splits do not fit the workgroup map (use t2s_gemm_wgrad_splits)", T, splits);
  static LdsFlags done = {};
  int rc = 0;
  if ((rc = g_reserve_lds(gemm_tn_bf16_kernel, done, "gemm_wgrad"))) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3((unsigned)(G_XCDS * c)), dim3(512), G_SMEM, st, p);
  T2S_CHECK_LAUNCH("gemm_wgrad");
  const int64_t n4 = (int64_t)n_out * n_in / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)slabs, dw, n4, splits, accumulate);
  T2S_CHECK_LAUNCH("gemm_wgrad reduce");
  return 0;
}
