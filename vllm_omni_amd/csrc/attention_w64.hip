// Flash attention forward, 64 queries per wave, ONE wave per SIMD: the large-batch production kernel (gfx950 only).
//
// flash_attn_fwd_pipe_kernel (attention.hip) gives each wave 32 queries and runs two waves per SIMD; it is bound by per-wave
// instruction issue (~370 issued instructions per wave and KV tile around 32 MFMAs).  Here a workgroup is 4 waves
// = one 256-query block of one (item, head); a wave owns 64 queries (two 32-query blocks) and the whole 512-entry register
// file of its SIMD, so every K / V^T fragment read from LDS feeds TWO MFMAs and the LDS-DMA per query halves:
//
//   AGPRs (named literally in the asm statements, never touched by the compiler):
//     a[0:127]    O^T accumulators  o[bq][d]  (bq = query block 0/1, d = 32-wide slice of head_dim)  = 8 x 16
//     a[128:191]  Q~ fragments      qf[bq][ks] (ks = 16-wide k-step of head_dim)                     = 16 x 4
//     a[192:255]  K fragments of ONE whole tile, kf[j][ks] (j = 32-key block)                        = 16 x 4
//   arch VGPRs (compiler-allocated): two S tiles (2 x 64), the packed P tile (32), the -max splats (32), V^T fragments in
//   flight (16), addresses.
//
// MFMA operands: S^T = K Q~^T takes A and B from AGPRs and C / D in VGPRs (the first MFMA of a chain takes C = -m_run, so the
// accumulator IS the exp2 argument: OMNI_ATTN_BAKE of attention.hip); O^T += V^T P^T takes A (V^T fragment, tr-read from LDS)
// and B (P^T = the exp'ed S registers, packed) from VGPRs and C / D in AGPRs.
//
// Per KV tile t a wave runs two phases of 32 MFMAs:
//   P1:  S(t+1) = K(t+1) Q~^T  ||  exp2 / pack / row-sum of S(t)                       (5 VALU per MFMA: the loaded phase)
//   P2:  O += V(t)^T P(t)^T    ||  row max of S(t+1)  ||  V(t) tr-reads (3 fragments ahead)  ||  K(t+2) fragments -> AGPRs
//                              ||  LDS-DMA issue for K(t+4), V(t+2)   (a piece costs its wave ~50 issue cycles: measured 11 %
//                                  of the kernel when issued inside P1, whose MFMA gaps are already full)
// and ONE barrier.  LDS: 3 K slots + 3 V slots of 16 KiB (images as in attention.hip: K rows XOR-swizzled, V in tr-read blocks).
//   K(j) lives in K slot j % 3: DMA'd in P2 of iteration j-4, retired by the COUNTED vmcnt(8) at the end of iteration j-3 (the 8
//        pieces of iteration j-3 stay in flight across the barrier), read into AGPRs in P2 of iteration j-2, multiplied in P1 of
//        iteration j-1; its slot takes K(j+3) in P2 of iteration j-1.
//   V(j) lives in V slot j % 3: DMA'd in P2 of iteration j-2, retired at the end of iteration j-1, read in P2 of iteration j; its
//        slot takes V(j+3) in P2 of iteration j+1.
// The slot of a tile is a run-time value (a period of 3 against the S ping-pong's period of 2 would need a 6-fold unroll):
// the nine fragment base addresses are re-based once per tile (9 VALU).
#include <atomic>
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

constexpr int DH = 128;
constexpr int KVBLK = 64;
constexpr int TILE_BYTES = KVBLK * DH * 2;        // 16 KiB
constexpr int NSLOT = 3;                          // K ring and V ring: 3 slots of 16 KiB each
constexpr int K_SLOT0 = 0, V_SLOT0 = NSLOT * TILE_BYTES;
constexpr int LDS_BYTES = 2 * NSLOT * TILE_BYTES; // 96 KiB
constexpr int QBLK = 256;                         // queries per workgroup

constexpr int A_O = 0, A_Q = 128, A_K = 192;

#ifndef OMNI_W64_P2SPLIT
#define OMNI_W64_P2SPLIT 1      // P2: the fillers of a step are split between its two MFMAs (0: both MFMAs back to back)
#endif
#ifndef OMNI_W64_KREAD_STEPS
#define OMNI_W64_KREAD_STEPS 8  // the 16 K(t+2) fragment reads are spread over the first N PV steps of P2 (16, 8 or 4)
#endif
#ifndef OMNI_W64_DMA_STEP0
#define OMNI_W64_DMA_STEP0 0    // the 8 LDS-DMA pieces of a tile are issued behind PV steps STEP0 .. STEP0+7 of P2
#endif
#ifndef OMNI_W64_HOISTV
#define OMNI_W64_HOISTV 1       // dev bisect knobs (all 1 in production)
#endif
#ifndef OMNI_W64_NEWDMA
#define OMNI_W64_NEWDMA 1
#endif
#ifndef OMNI_W64_FUSEDMAX
#define OMNI_W64_FUSEDMAX 1
#endif
#ifndef OMNI_W64_FINP2
#define OMNI_W64_FINP2 1
#endif
#ifndef OMNI_W64_EARLYDEC
#define OMNI_W64_EARLYDEC 1
#endif
#ifndef OMNI_W64_XHALF_IN_P2
#define OMNI_W64_XHALF_IN_P2 1
#endif
#ifndef OMNI_W64_IDLE_WAVES
#define OMNI_W64_IDLE_WAVES 1   // waves without query rows skip the arithmetic (0: they compute on clamped rows, as before)
#endif
#ifndef OMNI_W64_ABL
#define OMNI_W64_ABL 0          // dev-only timing ablations (WRONG results): 1 no DMA, 2 no end-of-tile wait + barrier, 4 no exp,
#endif                          // 8 no LDS fragment reads, 16 no row max, 32 no MFMA, 64 no K DMA pieces (V only), 128 no K fragment reads
#ifndef OMNI_DEV
// The knobs above exist for -DOMNI_DEV variant builds (tools/build_variants.sh).  A PRODUCT build must carry the measured
// production values: anything else on the command line is a build error, not a silently different kernel (round-5 verdict nit 14).
static_assert(OMNI_W64_P2SPLIT == 1 && OMNI_W64_KREAD_STEPS == 8 && OMNI_W64_DMA_STEP0 == 0 && OMNI_W64_HOISTV == 1 &&
                  OMNI_W64_NEWDMA == 1 && OMNI_W64_FUSEDMAX == 1 && OMNI_W64_FINP2 == 1 && OMNI_W64_EARLYDEC == 1 &&
                  OMNI_W64_XHALF_IN_P2 == 1 && OMNI_W64_IDLE_WAVES == 1 && OMNI_W64_ABL == 0,
              "attention_w64.hip: tuning knobs differ from their production values in a product (non -DOMNI_DEV) build");
#endif

// LDS issue order of P2 (LDS operations return in order, so a counted lgkmcnt retires exactly the reads a step needs):
//   VREAD(0) VREAD(1) VREAD(2) | step f: [wait] MFMAs, VREAD(f+3) (2 ops), KREADs of step f (16 / KREAD_STEPS ops, f < KREAD_STEPS)
constexpr int kreads_at(int f) { return (OMNI_W64_ABL & 128) ? 0 : f < OMNI_W64_KREAD_STEPS ? 16 / OMNI_W64_KREAD_STEPS : 0; }
constexpr int lds_ops_allowed_at(int f) {   // operations issued after VREAD(f) and before step f's wait
  int issued = 6, after_vf = f < 3 ? 2 * (f + 1) : 0;
  for (int s = 0; s < f; ++s) {
    if (s + 3 < 16) { issued += 2; if (s + 3 == f) after_vf = issued; }
    issued += kreads_at(s);
  }
  const int n = issued - after_vf;
  return n > 15 ? 15 : n;                   // lgkmcnt is a 4-bit counter; waiting for fewer is only conservative
}
static_assert(lds_ops_allowed_at(0) == 4, "VREAD(1), VREAD(2) may stay in flight at step 0");

// ---- asm statements.  hipcc schedules each as one opaque instruction and inserts no hazard padding inside: every wait state
// a statement needs is written in its string (cdna_hip_programming.md 5.7).
template <int KA, int QA, bool FRESH_C>   // first MFMA of an S chain: D <- A(K frag) x B(Q frag) + C, C = the -max splat.
OMNI_DEVINL void mfma_qk_first(f32x16_t& d, const f32x16_t& c) {
  // FRESH_C: the splat may have been (re)materialised by VALU moves right in front of this statement (prologue: the compiler
  // places the zero splat of tile 0 there) — a VALU-written SrcC needs wait states and they must sit INSIDE the statement.
  // In the loop the splat is long-lived (rewritten only by the rescale path, which pads itself): no wait state, no issue slot.
  if constexpr (FRESH_C)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, a[%c2:%c3], a[%c4:%c5], %1"
                 : "=&v"(d) : "v"(c), "i"(KA), "i"(KA + 3), "i"(QA), "i"(QA + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c2:%c3], a[%c4:%c5], %1"
                 : "=&v"(d) : "v"(c), "i"(KA), "i"(KA + 3), "i"(QA), "i"(QA + 3));
}
template <int KA, int QA>
OMNI_DEVINL void mfma_qk_acc(f32x16_t& d) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(d) : "i"(KA), "i"(KA + 3), "i"(QA), "i"(QA + 3));
}
template <int OA>           // O^T[bq][d] += V^T frag x P^T frag   (accumulator in AGPRs)
OMNI_DEVINL void mfma_pv(const u32x4_t& vf, const u32x4_t& pf) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vf), "v"(pf), "i"(OA), "i"(OA + 15));
}
template <int KA, int OFF>  // one K fragment (16 B per lane) LDS -> AGPRs
OMNI_DEVINL void kread(uint32_t addr) {
  asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "i"(KA), "i"(KA + 3), "i"(OFF));
}
template <int OFF>
OMNI_DEVINL u32x2_t vread8(uint32_t addr) {   // ds_read_b64_tr_b16: hardware 4x4 transpose read (half of a V^T fragment)
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
template <int A>
OMNI_DEVINL void agpr_write(uint32_t v) { asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(v), "i"(A)); }
template <int A>
OMNI_DEVINL void agpr_zero() { asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"i"(A)); }
template <int A>
OMNI_DEVINL float agpr_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(A));
  return v;
}
template <int A>
OMNI_DEVINL void agpr_scale(float alpha) {
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a[%c2], %0"
               : "=&v"(t) : "v"(alpha), "i"(A));
}
template <int BASE, int... I>
OMNI_DEVINL void agpr_zero_range(std::integer_sequence<int, I...>) { (agpr_zero<BASE + I>(), ...); }
template <int BASE, int... I>
OMNI_DEVINL void agpr_scale_range(float alpha, std::integer_sequence<int, I...>) { (agpr_scale<BASE + I>(alpha), ...); }
// a PV MFMA's D reaches a non-MFMA reader only after its passes have drained: software wait states, no interlock
OMNI_DEVINL void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
// the same, pinned in front of plain VALU code that touches an S tile right behind the MFMAs that produced it
OMNI_DEVINL void mfma_drain_s(f32x16_t (&S)[2][2]) {
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[1][0]), "+v"(S[1][1]));
}

OMNI_DEVINL float max3(float a, float b, float c) {   // one instruction; fmaxf() on asm outputs gets a canonicalising v_max first
  float r;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// four v_max3 of the two row-max chains, interleaved, in ONE statement: the compiler pads back-to-back dependent asm VALU
// statements with s_nop (6 issue slots for 4 instructions); inside a statement nothing is inserted and none is needed
OMNI_DEVINL void max3x4(float& m0, float& m1, float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3) {
  asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %8, %9"
               : "+v"(m0), "+v"(m1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}
OMNI_DEVINL void max3x4_first(float& m0, float& m1, float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3) {
  asm volatile("v_max3_f32 %0, %2, %3, %4\n\tv_max3_f32 %1, %6, %7, %8\n\tv_max_f32 %0, %0, %5\n\tv_max_f32 %1, %1, %9"
               : "=&v"(m0), "=&v"(m1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}
OMNI_DEVINL float xhalf_max(float x) {          // max(x, x of the lane 32 away): one statement, 5 issue slots
  float a = x, b;
  asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "=&v"(b));
  return a;
}
OMNI_DEVINL float xhalf_sum(float x) {
  uint32_t a = __builtin_bit_cast(uint32_t, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
// LDS-DMA through a buffer descriptor: per-lane byte offset (offen) + uniform soffset; M0 = LDS byte address of the 1-KiB
// piece (the hardware adds lane * 16).  Rows past the end of the sequence are outside the descriptor's range and land as
// ZEROS: no clamp, no branch (those keys are masked to -inf, their V rows meet P = 0).  No instruction offset: it would
// also move the LDS address.
OMNI_DEVINL void dma16(const u32x4_t& srd, uint32_t lane_off, uint32_t soff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               ::"s"(lds_addr), "v"(lane_off), "s"(srd), "s"(soff) : "memory");
}
// the in-loop form: M0 and the source offset are formed by two SALU adds inside the statement (the second one is the wait
// state the M0 write needs before the DMA): 3 issue slots per piece instead of 5
template <int LDS_IMM, int SOFF_IMM>
OMNI_DEVINL void dma16_at(const u32x4_t& srd, uint32_t lane_off, uint32_t soff_base, uint32_t lds_base) {
  uint32_t tmp;
  asm volatile("s_add_u32 m0, %1, %c5\n\ts_add_u32 %0, %2, %c6\n\tbuffer_load_dwordx4 %3, %4, %0 offen lds"
               : "=&s"(tmp) : "s"(lds_base), "s"(soff_base), "v"(lane_off), "s"(srd), "i"(LDS_IMM), "i"(SOFF_IMM) : "memory", "scc");
}
template <int LDS_IMM>      // the same with a run-time (uniform) source addend
OMNI_DEVINL void dma16_at(const u32x4_t& srd, uint32_t lane_off, uint32_t soff_base, uint32_t soff_add, uint32_t lds_base) {
  uint32_t tmp;
  asm volatile("s_add_u32 m0, %1, %c6\n\ts_add_u32 %0, %2, %5\n\tbuffer_load_dwordx4 %3, %4, %0 offen lds"
               : "=&s"(tmp) : "s"(lds_base), "s"(soff_base), "v"(lane_off), "s"(srd), "s"(soff_add), "i"(LDS_IMM) : "memory", "scc");
}
OMNI_DEVINL u32x4_t make_srd(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  u32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);     // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

template <int N>
using ic = std::integral_constant<int, N>;

// Split mode (nsplit > 1; host: omni_internal_flash_attn_w64): the grid lists the FULL 256-query blocks first (qfull per (item,
// head), XCD-aware order as before) and behind them, for the item's short LAST block (<= 64 query rows: 4160 = 16 x 256 + 64, 16448 =
// 64 x 256 + 64 — one wave of four has rows), nsplit workgroups per (item, head) that each visit 1 / nsplit of the key tiles and
// leave an UN-normalised partial (O, running max, row sum: fp32) in `part_o` / `part_ml`; attn_split_combine_kernel merges them.
// The short block otherwise costs a whole workgroup-time with three idle waves, dispatched last: at 2 x 24 heads x 65 blocks
// (one 2048^2 request) the 13th round of the grid holds 48 workgroups on 256 CUs.
__global__ __launch_bounds__(256, 1) void flash_attn_fwd_w64_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    uint16_t* __restrict__ out, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
    const int32_t* __restrict__ cu_seqlens, int n_heads_total, int H, float scale_log2e, int out_k32_rows,
    const int32_t* __restrict__ item_skip, int q_prescaled, int qfull, int nsplit, float* __restrict__ part_o,
    float* __restrict__ part_ml) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  // XCD-aware head-major block order (attention.hip): block b runs on XCD b % 8; every XCD gets a contiguous range of the
  // (item*head, q-block) list, so the workgroups resident on an XCD stream the same few heads' K / V through its L2
  int hb, qb, sp = 0;
  const int nfull = n_heads_total * qfull;          // == gridDim.x without split mode
  const bool split = (int)blockIdx.x >= nfull;
  if (!split) {
    const int nwg = nfull, bid = blockIdx.x, qblocks = qfull;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    hb = lid / qblocks;
    qb = lid - hb * qblocks;
  } else {
    const int r = (int)blockIdx.x - nfull;
    hb = r / nsplit;
    sp = r - hb * nsplit;
    qb = qfull;
  }
  const int b = hb / H, h = hb - b * H;
  if (item_skip && item_skip[b]) return;
  const int seq_start = cu_seqlens[b];
  const int seq_len = cu_seqlens[b + 1] - seq_start;
  if (qb * QBLK >= seq_len) return;

  asm volatile(OMNI_OWNS_AGPRS ::: OMNI_ALL_AGPRS);   // allocate a[0:255]

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // key tiles of this workgroup: all of them, or (split mode) the sp-th of nsplit contiguous ranges.  Everything below indexes
  // tiles and keys RELATIVE to the range: the buffer descriptors start at its first key and end at its last.
  const int ntiles_all = (seq_len + KVBLK - 1) / KVBLK;
  const int t_first = split ? (int)((long)sp * ntiles_all / nsplit) : 0;
  const int t_last = split ? (int)((long)(sp + 1) * ntiles_all / nsplit) : ntiles_all;
  const int ntiles = t_last - t_first;
  const int kv_row0 = t_first * KVBLK;
  const int kv_rows = min(seq_len, t_last * KVBLK) - kv_row0;     // keys of the range (the last range ends with the sequence)
  if (ntiles <= 0) {                                // fewer tiles than splits: an empty partial (l = 0; the combine skips it)
    if (wave == 0 && lane < 64) {
      float* ml = part_ml + ((int64_t)(hb * nsplit + sp) * 64 + lane) * 2;
      ml[0] = -INFINITY; ml[1] = 0.0f;
    }
    return;
  }

  // ---- LDS-DMA sources.  Piece P = wave + 4i (1 KiB of a tile image), lane L -> byte 16 L of the piece.
  //  K piece P: key = 4P + (L>>4) = [4 wave + (L>>4)] + 16 i; LDS chunk L&15 holds logical chunk (L&15) ^ (key&15) — the same
  //             for every i, so the four pieces differ by a uniform 16-row stride in soffset;
  //  V piece P = (dblk = i, key group = wave): key = 16 wave + 4 (L>>4) + ((L>>2)&3), logical chunk 4i + (L&3): the four
  //             pieces differ by 64 B in soffset.
  const u32x4_t k_srd = make_srd(k + (int64_t)(seq_start + kv_row0) * ldk + h * DH, (uint32_t)((int64_t)(kv_rows - 1) * ldk * 2 + DH * 2));
  const u32x4_t v_srd = make_srd(v + (int64_t)(seq_start + kv_row0) * ldv + h * DH, (uint32_t)((int64_t)(kv_rows - 1) * ldv * 2 + DH * 2));
  const int k_key0 = 4 * wave + (lane >> 4);
  const int v_key = 16 * wave + 4 * (lane >> 4) + ((lane >> 2) & 3);
  const uint32_t k_src = (uint32_t)(k_key0 * ldk * 2) + (uint32_t)(((lane & 15) ^ (k_key0 & 15)) * 16);
  const uint32_t v_src = (uint32_t)(v_key * ldv * 2) + (uint32_t)((lane & 3) * 16);
  const uint32_t k_tile_stride = (uint32_t)(KVBLK * ldk * 2), v_tile_stride = (uint32_t)(KVBLK * ldv * 2);
  const uint32_t k_piece_stride = (uint32_t)(16 * ldk * 2);
  // a tile index past the last tile is fine: every row of it is out of the descriptor's range (zeros)
  auto issue_K_piece = [&](int t, int slot, auto ii) {
    constexpr int i = decltype(ii)::value;
    dma16(k_srd, k_src, (uint32_t)t * k_tile_stride + i * k_piece_stride, lds0 + K_SLOT0 + slot * TILE_BYTES + (wave + 4 * i) * 1024);
  };
  auto issue_V_piece = [&](int t, int slot, auto ii) {
    constexpr int i = decltype(ii)::value;
    dma16(v_srd, v_src, (uint32_t)t * v_tile_stride + i * 64, lds0 + V_SLOT0 + slot * TILE_BYTES + (wave + 4 * i) * 1024);
  };
  auto issue_K = [&](int t, int slot) {
    issue_K_piece(t, slot, ic<0>{}); issue_K_piece(t, slot, ic<1>{}); issue_K_piece(t, slot, ic<2>{}); issue_K_piece(t, slot, ic<3>{});
  };
  auto issue_V = [&](int t, int slot) {
    issue_V_piece(t, slot, ic<0>{}); issue_V_piece(t, slot, ic<1>{}); issue_V_piece(t, slot, ic<2>{}); issue_V_piece(t, slot, ic<3>{});
  };

  // ---- prologue DMA first (it flies while Q is fetched)
  issue_K(0, 0);
  issue_K(1, 1);
  issue_K(2, 2);
  issue_V(0, 0);
  issue_V(1, 1);

  // ---- a wave WITHOUT query rows (the item's last q-block: 4160 = 16 x 256 + 64 leaves three of four waves empty in one
  // block of seventeen) only feeds the rings: the same DMA pieces, counted waits and barriers as the loop below, no MFMA, no
  // exp — the matrix pipe's energy for rows nobody stores is what the power-capped part gives back as clock
  if (OMNI_W64_IDLE_WAVES && qb * QBLK + wave * 64 >= seq_len) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // (tiles 0 .. 2 published)
    __syncthreads();                                  // (every wave has K(0) in registers)
    issue_K(3, 0);
    __syncthreads();                                  // (end of the prologue)
    int st = 0;
    for (int t = 0; t < ntiles; ++t) {
      const int s1 = st == 2 ? 0 : st + 1, s2 = st == 0 ? 2 : st - 1;
      issue_V(t + 2, s2);
      issue_K(t + 4, s1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __syncthreads();
      st = s1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- Q~ fragments (B operand: lane holds query l31, d = ks*16 + hi*8 .. +8) -> a[128:191]; O <- 0
  {
    bf16x8_t qf[2][8];
    const float qs = q_prescaled ? 1.0f : scale_log2e;
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) {
      const int qrow = min(qb * QBLK + wave * 64 + bq * 32 + l31, seq_len - 1);
      const uint16_t* qp = q + (int64_t)(seq_start + qrow) * ldq + h * DH + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) qf[bq][ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
    }
    uint32_t qw[64];
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        u32x4_t w = __builtin_bit_cast(u32x4_t, qf[bq][ks]);
        // the reference's un-scaled q: Q~ = bf16(q * scale * log2 e) (one extra rounding; attention.hip); a pre-scaled q is
        // multiplied by 1.0f (exact)
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf16x2(bf16_lo(w[e]) * qs, bf16_hi(w[e]) * qs);
#pragma unroll
        for (int e = 0; e < 4; ++e) qw[(bq * 8 + ks) * 4 + e] = w[e];
      }
    [&]<int... I>(std::integer_sequence<int, I...>) { (agpr_write<A_Q + I>(qw[I]), ...); }(std::make_integer_sequence<int, 64>{});
    agpr_zero_range<A_O>(std::make_integer_sequence<int, 128>{});
  }

  // ---- fragment read addresses (attention.hip).  K: row key = j*32 + l31, chunk (ks*2 + hi) ^ (key & 15)
  uint32_t k_addr[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_addr[ks] = lds0 + K_SLOT0 + l31 * 256 + ((((uint32_t)(ks * 2 + hi)) ^ (l31 & 15)) << 4);
  const uint32_t v_addr = lds0 + V_SLOT0 + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8 + hi * 256;

  float negm[2];
  f32x16_t negm16[2];
#pragma unroll
  for (int bq = 0; bq < 2; ++bq)
#pragma unroll
    for (int i = 0; i < 16; ++i) negm16[bq][i] = 0.0f;

  // all 16 K fragments of the tile in K slot `slot` -> a[192:255]   (fragment f = j*8 + ks)
  auto kread_one = [&](auto ff, const uint32_t (&ka)[8]) {
    constexpr int f = decltype(ff)::value;
    kread<A_K + f * 4, (f >> 3) * 32 * 256>(ka[f & 7]);
  };
  auto kread_all = [&](int slot) {
    uint32_t ka[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ka[ks] = k_addr[ks] + slot * TILE_BYTES;
    [&]<int... F>(std::integer_sequence<int, F...>) { (kread_one(ic<F>{}, ka), ...); }(std::make_integer_sequence<int, 16>{});
  };

  // one QK^T MFMA step i = 0..31: j = i >> 4, ks = (i >> 1) & 7, bq = i & 1
  auto qk_step = [&](auto ii, f32x16_t (&SN)[2][2], auto fresh_c) {
    constexpr int i = decltype(ii)::value, j = i >> 4, ks = (i >> 1) & 7, bq = i & 1;
    constexpr bool FRESH = decltype(fresh_c)::value;
    if constexpr (OMNI_W64_ABL & 32) {
      if constexpr (ks == 0) SN[bq][j] = negm16[bq];
      f32x16_t& sref = SN[bq][j];
      asm volatile("" : "+v"(sref));
    }
    else if constexpr (ks == 0) mfma_qk_first<A_K + (j * 8 + ks) * 4, A_Q + (bq * 8 + ks) * 4, FRESH>(SN[bq][j], negm16[bq]);
    else mfma_qk_acc<A_K + (j * 8 + ks) * 4, A_Q + (bq * 8 + ks) * 4>(SN[bq][j]);
  };
  auto mask_tail = [&](f32x16_t (&S)[2][2], int kv0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kv0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key >= kv_rows) { S[0][j][r] = -INFINITY; S[1][j][r] = -INFINITY; }
      }
  };
  auto row_max = [&](f32x16_t (&S)[2][2], float (&mx)[2]) {
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) {
      float m = S[bq][0][0];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        m = max3(m, S[bq][c >> 2][(c & 3) * 4 + 0], S[bq][c >> 2][(c & 3) * 4 + 1]);
        m = max3(m, S[bq][c >> 2][(c & 3) * 4 + 2], S[bq][c >> 2][(c & 3) * 4 + 3]);
      }
      mx[bq] = xhalf_max(m);
    }
  };

  // ---- prologue: K(0) -> AGPRs, S(0), K(2) into the freed slot, K(1) -> AGPRs, row max of S(0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  kread_all(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();                                  // every wave has K(0) in registers: its slot can take K(3)
  issue_K(3, 0);
  f32x16_t sA[2][2], sB[2][2];
  float mxA[2], mxB[2] = {0.0f, 0.0f};
  [&]<int... I>(std::integer_sequence<int, I...>) { (qk_step(ic<I>{}, sA, std::true_type{}), ...); }(std::make_integer_sequence<int, 32>{});
  __builtin_amdgcn_sched_barrier(0);
  kread_all(1);
  mfma_drain_s(sA);                                 // S(0) is read by VALU right away here (in the loop it is not)
  if (KVBLK > kv_rows) mask_tail(sA, 0);
  row_max(sA, mxA);
  // tile 0 opened its chains with C = 0: its row max becomes the first running max here (O = l = 0: nothing to rescale), also
  // when every score of the tile is far below zero; the loop then sees a tile whose max is exactly at the reference point
#pragma unroll
  for (int bq = 0; bq < 2; ++bq) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) sA[bq][j][i] -= mxA[bq];
    negm[bq] = -mxA[bq];
#pragma unroll
    for (int i = 0; i < 16; ++i) negm16[bq][i] = negm[bq];
    mxA[bq] = 0.0f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // K(1) in AGPRs (K(3) may stay in flight: retired in iteration 0)
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();

  // One iteration.  SC = S(t) (complete; row max mxc), SN receives S(t+1).  HAS_NEXT is compile-time (the last tile's body has
  // no QK^T).  Everything that is not an MFMA sits in an MFMA's shadow; what used to run between the phases and between the
  // iterations (round-3 timeline: ~45 + ~30 issue slots with the matrix pipe idle, plus an exposed LDS read latency) is now
  // loop-carried and produced inside P2:
  //   * the first three V(t) fragments are read right behind the barrier that published the tile (before P1), not at P2's door;
  //   * the fragment base addresses of the NEXT tile's slots are formed in P2's last steps;
  //   * the row max of S(t+1) is complete by step 9, its cross-half exchange and the rescale decision follow in steps 10-12:
  //     the next iteration opens with one scalar branch;
  //   * the row sums are two persistent accumulator pairs (no per-tile zeroing / folding), the last two pack/sum chunks of P1
  //     ride in P2's first two gaps (their P words are first used by step 12).
  int slot_t = 0;                                   // t % 3, maintained by the loop
  float lsum[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};  // row sums: two interleaved chains per query block, never reset
  uint32_t va = v_addr;                             // V(t) fragment base:   V slot t % 3
  uint32_t ka[8];                                   // K(t+2) fragment bases: K slot (t+2) % 3
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) ka[ks] = k_addr[ks] + 2 * TILE_BYTES;
  u32x2_t vlo[4], vhi[4];                           // V^T fragments in flight
  bool rescale = false;                             // decided in P2 of the previous iteration (tile 0 is at its own max)
  uint32_t kps[4];                                  // K piece strides as uniform values
#pragma unroll
  for (int i = 0; i < 4; ++i) kps[i] = __builtin_amdgcn_readfirstlane(i * k_piece_stride);

  auto vread = [&](auto ff) {
    constexpr int f = decltype(ff)::value;
    constexpr int off = (f & 3) * 4096 + (((f >> 3) * 8 + ((f >> 2) & 1) * 4) * 256);
    if constexpr (OMNI_W64_ABL & 8) {
      const uint32_t va_ = va;
      u32x2_t a_, b_;
      asm volatile("" : "=v"(a_) : "v"(va_));
      asm volatile("" : "=v"(b_) : "v"(va_));
      vlo[f & 3] = a_; vhi[f & 3] = b_;
    } else {
      vlo[f & 3] = vread8<off>(va);
      vhi[f & 3] = vread8<off + 512>(va);
    }
  };

  auto iteration = [&](auto has_next_c, int t, f32x16_t (&SC)[2][2], f32x16_t (&SN)[2][2], float (&mxc)[2],
                       float (&mxn)[2]) {
    constexpr bool HAS_NEXT = decltype(has_next_c)::value;
    const int s1 = slot_t == 2 ? 0 : slot_t + 1, s2 = slot_t == 0 ? 2 : slot_t - 1;   // (t+1) % 3, (t+2) % 3
    // DMA targets of this iteration: V(t+2) -> V slot (t+2) % 3 [held V(t-1)], K(t+4) -> K slot (t+1) % 3 [held K(t+1)]
    const uint32_t v_lds = lds0 + V_SLOT0 + s2 * TILE_BYTES + wave * 1024, k_lds = lds0 + K_SLOT0 + s1 * TILE_BYTES + wave * 1024;
    const uint32_t v_soff = (uint32_t)(t + 2) * v_tile_stride, k_soff = (uint32_t)(t + 4) * k_tile_stride;

    if constexpr (OMNI_W64_HOISTV) { vread(ic<0>{}); vread(ic<1>{}); vread(ic<2>{}); }   // V(t) landed before the barrier behind us

    // ---- defer-max (attention.hip OMNI_ATTN_BAKE): SC already is s~ - m~; rescale only when a row max exceeds the threshold
    // (rare after the first tiles; the wave-uniform decision was taken in P2 of the previous iteration)
    constexpr float DEFER = 6.0f;
    u32x4_t pf[2][2][2];                            // packed P^T: [bq][j][16-key half] = one MFMA B operand
    if constexpr (!OMNI_W64_EARLYDEC) rescale = !__all(max3(mxc[0], mxc[0], mxc[1]) <= DEFER);
    if (rescale) {
#pragma unroll
      for (int bq = 0; bq < 2; ++bq) {
        if (!__all(mxc[bq] <= DEFER)) {
          const float d = fmaxf(mxc[bq], 0.0f);
          const float alpha = __builtin_amdgcn_exp2f(-d);
          lsum[bq][0] *= alpha;
          lsum[bq][1] *= alpha;
          mfma_drain();
          if (bq == 0) agpr_scale_range<A_O>(alpha, std::make_integer_sequence<int, 64>{});
          else agpr_scale_range<A_O + 64>(alpha, std::make_integer_sequence<int, 64>{});
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) SC[bq][j][i] -= d;
#pragma unroll
          for (int i = 0; i < 16; ++i) negm16[bq][i] -= d;
        }
      }
      asm volatile("s_nop 4" ::: "memory");         // the -max splats were just written by VALU: wait states before an MFMA reads them as C
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- P1: exp chunk c = 0..31 <-> (bq = c & 1, e = c >> 1): S elements 2e, 2e+1 of SC[bq] -> one packed word of P.
    // The pack / sum of chunk c is issued behind the exps of chunk c+1 (a v_exp result needs a wait state before a VALU read).
    float pend0[2], pend1[2];
    auto exp_finish = [&](auto cc) {
      constexpr int c = decltype(cc)::value, bq = c & 1, e = c >> 1;
      lsum[bq][0] += pend0[bq];
      lsum[bq][1] += pend1[bq];
      uint32_t pk = pack_bf16x2(pend0[bq], pend1[bq]);
      asm volatile("" : "+v"(pk), "+v"(lsum[bq][0]), "+v"(lsum[bq][1]));      // pin: keep it between these two MFMAs
      pf[bq][e >> 3][(e >> 2) & 1][e & 3] = pk;
    };
    auto exp_chunk = [&](auto cc) {
      constexpr int c = decltype(cc)::value, bq = c & 1, e = c >> 1;
      float e0 = (OMNI_W64_ABL & 4) ? SC[bq][e >> 3][(2 * e) & 15] : __builtin_amdgcn_exp2f(SC[bq][e >> 3][(2 * e) & 15]);
      float e1 = (OMNI_W64_ABL & 4) ? SC[bq][e >> 3][((2 * e) & 15) + 1] : __builtin_amdgcn_exp2f(SC[bq][e >> 3][((2 * e) & 15) + 1]);
      asm volatile("" : "+v"(e0), "+v"(e1));
      if constexpr (c >= 2) exp_finish(ic<c - 2>{});
      pend0[bq] = e0; pend1[bq] = e1;
    };
    auto p1_step = [&](auto ii) {
      if constexpr (HAS_NEXT) {
        qk_step(ii, SN, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
      }
      exp_chunk(ii);
      __builtin_amdgcn_sched_barrier(0);
    };
    __builtin_amdgcn_s_setprio(1);
    [&]<int... I>(std::integer_sequence<int, I...>) { (p1_step(ic<I>{}), ...); }(std::make_integer_sequence<int, 32>{});
    if constexpr (!OMNI_W64_FINP2) { exp_finish(ic<30>{}); exp_finish(ic<31>{}); }
    if (HAS_NEXT && (t + 2) * KVBLK > kv_rows) { mfma_drain_s(SN); mask_tail(SN, (t + 1) * KVBLK); }
    __builtin_amdgcn_sched_barrier(0);

    // ---- P2: O^T += V(t)^T P(t)^T.  PV step f = 0..15 <-> (j = f >> 3, half = (f >> 2) & 1, d = f & 3): ONE V^T fragment
    // (two tr-reads), two MFMAs (bq = 0, 1).  Behind the FIRST MFMA of a step (nothing there may overwrite its V fragment):
    // steps 0-1 the last two pack/sum chunks, 2-9 the row max of S(t+1), 10-11 its cross-half exchange, 12 the decision.
    // Behind the SECOND: V fragment f+3, K(t+2) fragments -> AGPRs (after P1 nothing reads a[192:255]), one LDS-DMA piece, and
    // in steps 13-15 the next tile's fragment bases.
    // LDS ops return in order: at step f the reads issued after VREAD(f) are VREAD(f+1), VREAD(f+2) and up to three KREADs.
    if constexpr (!OMNI_W64_HOISTV) { vread(ic<0>{}); vread(ic<1>{}); vread(ic<2>{}); }
    float mx[2] = {0.0f, 0.0f};
    auto first_gap = [&](auto ff) {
      constexpr int f = decltype(ff)::value, c = f - 2;
      if constexpr (f == 0 && OMNI_W64_FINP2) exp_finish(ic<30>{});
      if constexpr (f == 1 && OMNI_W64_FINP2) exp_finish(ic<31>{});
      if constexpr (HAS_NEXT && !(OMNI_W64_ABL & 16)) {
        if constexpr (!OMNI_W64_FUSEDMAX) {
          if constexpr (c == 0) { mx[0] = SN[0][0][0]; mx[1] = SN[1][0][0]; }
          if constexpr (c >= 0 && c < 8) {
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) {
              mx[bq] = max3(mx[bq], SN[bq][c >> 2][(c & 3) * 4 + 0], SN[bq][c >> 2][(c & 3) * 4 + 1]);
              mx[bq] = max3(mx[bq], SN[bq][c >> 2][(c & 3) * 4 + 2], SN[bq][c >> 2][(c & 3) * 4 + 3]);
            }
          }
        } else if constexpr (c == 0)
          max3x4_first(mx[0], mx[1], SN[0][0][0], SN[0][0][1], SN[0][0][2], SN[0][0][3], SN[1][0][0], SN[1][0][1], SN[1][0][2], SN[1][0][3]);
        else if constexpr (c > 0 && c < 8)
          max3x4(mx[0], mx[1], SN[0][c >> 2][(c & 3) * 4 + 0], SN[0][c >> 2][(c & 3) * 4 + 1], SN[0][c >> 2][(c & 3) * 4 + 2],
                 SN[0][c >> 2][(c & 3) * 4 + 3], SN[1][c >> 2][(c & 3) * 4 + 0], SN[1][c >> 2][(c & 3) * 4 + 1],
                 SN[1][c >> 2][(c & 3) * 4 + 2], SN[1][c >> 2][(c & 3) * 4 + 3]);
        if constexpr (f == 10 && OMNI_W64_XHALF_IN_P2) mxn[0] = xhalf_max(mx[0]);
        if constexpr (f == 11 && OMNI_W64_XHALF_IN_P2) mxn[1] = xhalf_max(mx[1]);
        if constexpr (f == 12 && OMNI_W64_XHALF_IN_P2 && OMNI_W64_EARLYDEC) rescale = !__all(max3(mxn[0], mxn[0], mxn[1]) <= DEFER);
      }
    };
    auto kreads_of = [&](auto ff) {                 // K(t+2) (zeros past the last tile: unused)
      constexpr int f = decltype(ff)::value, n = kreads_at(f);
      if constexpr (!(OMNI_W64_ABL & (8 | 128)) && n > 0)
        [&]<int... J>(std::integer_sequence<int, J...>) { (kread_one(ic<f * n + J>{}, ka), ...); }(std::make_integer_sequence<int, n>{});
    };
    auto p2_step = [&](auto ff) {
      constexpr int f = decltype(ff)::value;
      if constexpr (!(OMNI_W64_ABL & 8)) asm volatile("s_waitcnt lgkmcnt(%c0)" ::"i"(lds_ops_allowed_at(f)) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      const u32x4_t w = {vlo[f & 3][0], vlo[f & 3][1], vhi[f & 3][0], vhi[f & 3][1]};
      const u32x4_t p0 = pf[0][f >> 3][(f >> 2) & 1], p1 = pf[1][f >> 3][(f >> 2) & 1];
      if constexpr (!(OMNI_W64_ABL & 32)) mfma_pv<A_O + (0 * 4 + (f & 3)) * 16>(w, p0);
      else asm volatile("" ::"v"(w), "v"(p0));
      __builtin_amdgcn_sched_barrier(0);
      first_gap(ff);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(OMNI_W64_ABL & 32)) mfma_pv<A_O + (1 * 4 + (f & 3)) * 16>(w, p1);
      else asm volatile("" ::"v"(w), "v"(p1));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(OMNI_W64_ABL & 1) && f >= OMNI_W64_DMA_STEP0 && f < OMNI_W64_DMA_STEP0 + 8) {
        constexpr int pc = f - OMNI_W64_DMA_STEP0;
        if constexpr (!OMNI_W64_NEWDMA) {
          if constexpr (pc < 4) issue_V_piece(t + 2, s2, ic<pc>{});
          else issue_K_piece(t + 4, s1, ic<pc - 4>{});
        } else if constexpr (pc < 4) dma16_at<pc * 4096, pc * 64>(v_srd, v_src, v_soff, v_lds);
        else if constexpr (!(OMNI_W64_ABL & 64)) dma16_at<(pc - 4) * 4096>(k_srd, k_src, k_soff, kps[pc - 4], k_lds);
      }
      if constexpr (f + 3 < 16) vread(ic<f + 3>{});
      kreads_of(ff);
      if constexpr (HAS_NEXT) {                     // next tile: V(t+1) in V slot (t+1) % 3, K(t+3) in K slot t % 3
        if constexpr (f == 13) va = v_addr + s1 * TILE_BYTES;     // (the last V(t) read was issued in step 12)
        if constexpr (f >= 13) {
          constexpr int k0 = (f - 13) * 3;
#pragma unroll
          for (int ks = k0; ks < (k0 + 3 < 8 ? k0 + 3 : 8); ++ks) ka[ks] = k_addr[ks] + slot_t * TILE_BYTES;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    [&]<int... F>(std::integer_sequence<int, F...>) { (p2_step(ic<F>{}), ...); }(std::make_integer_sequence<int, 16>{});
    __builtin_amdgcn_s_setprio(0);
    if constexpr (HAS_NEXT && !OMNI_W64_XHALF_IN_P2) {
      mxn[0] = xhalf_max(mx[0]);
      mxn[1] = xhalf_max(mx[1]);
      if constexpr (OMNI_W64_EARLYDEC) rescale = !__all(max3(mxn[0], mxn[0], mxn[1]) <= DEFER);
    }
    if constexpr (!(OMNI_W64_ABL & 2)) {
      // everything but this iteration's 8 pieces has landed: K(t+3), V(t+1) (issued one iteration ago); K(t+2) is in AGPRs
      asm volatile("s_waitcnt vmcnt(%c0)\n\ts_waitcnt lgkmcnt(0)" ::"i"((OMNI_W64_ABL & 1) ? 0 : (OMNI_W64_ABL & 64) ? 4 : 8) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    }
  };

  {
    using yes = std::true_type;
    using no = std::false_type;
    auto bump = [&] { slot_t = slot_t == 2 ? 0 : slot_t + 1; };
    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
      iteration(yes{}, t, sA, sB, mxA, mxB); bump();
      iteration(yes{}, t + 1, sB, sA, mxB, mxA); bump();
    }
    if (t + 1 < ntiles) {
      iteration(yes{}, t, sA, sB, mxA, mxB); bump();
      iteration(no{}, t + 1, sB, sA, mxB, mxA);
    } else {
      iteration(no{}, t, sA, sB, mxA, mxB);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // DMA pieces of tiles past the end (zeros) still target this block's LDS

  // ---- epilogue: O / l -> out.  o[bq][d][4 qd + j] = O[q][d*32 + qd*8 + hi*4 + j]
  mfma_drain();
  if (split) {
    // un-normalised partial of this key range: O (relative to the running max m~ of the range, exp2 domain), m~ and the row sum.
    // The short block has at most 64 rows: they all belong to wave 0 (the host enables split mode only then).
    if (wave == 0) {
#pragma unroll
      for (int bq = 0; bq < 2; ++bq) {
        const float l = xhalf_sum(lsum[bq][0] + lsum[bq][1]);
        float o[64];
        if (bq == 0) [&]<int... I>(std::integer_sequence<int, I...>) { ((o[I] = agpr_read<A_O + I>()), ...); }(std::make_integer_sequence<int, 64>{});
        else [&]<int... I>(std::integer_sequence<int, I...>) { ((o[I] = agpr_read<A_O + 64 + I>()), ...); }(std::make_integer_sequence<int, 64>{});
        const int64_t prow = (int64_t)(hb * nsplit + sp) * 64 + bq * 32 + l31;
        float* po = part_o + prow * DH + hi * 4;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const f32x4_t w = {o[d * 16 + qd * 4 + 0], o[d * 16 + qd * 4 + 1], o[d * 16 + qd * 4 + 2], o[d * 16 + qd * 4 + 3]};
            *reinterpret_cast<f32x4_t*>(po + d * 32 + qd * 8) = w;
          }
        if (hi == 0) {
          part_ml[prow * 2 + 0] = -negm16[bq][0];
          part_ml[prow * 2 + 1] = l;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int bq = 0; bq < 2; ++bq) {
    const float inv = 1.0f / xhalf_sum(lsum[bq][0] + lsum[bq][1]);
    const int qrow = qb * QBLK + wave * 64 + bq * 32 + l31;
    float o[64];
    if (bq == 0) [&]<int... I>(std::integer_sequence<int, I...>) { ((o[I] = agpr_read<A_O + I>()), ...); }(std::make_integer_sequence<int, 64>{});
    else [&]<int... I>(std::integer_sequence<int, I...>) { ((o[I] = agpr_read<A_O + 64 + I>()), ...); }(std::make_integer_sequence<int, 64>{});
    // A lane holds columns 8 qd + 4 hi .. +3 of its row, the other half-wave the neighbouring four: one v_permlane32_swap per
    // dword pairs (qd, qd+1) so that every lane owns 16 contiguous bytes — lanes 0-31 columns 16k .. 16k+7, lanes 32-63 columns
    // 16k+8 .. 16k+15 — and the row goes out as 8 x 16-B stores instead of 16 x 8-B (the store tail is issue-bound:
    // cdna_hip_programming.md T21).  All 64 lanes take part in the swaps; only the store is predicated.
    uint16_t* op = out_k32_rows ? out + ((int64_t)(h * 4) * out_k32_rows + seq_start + qrow) * 32 + hi * 8
                                : out + (int64_t)(seq_start + qrow) * ldo + h * DH + hi * 8;
    const int64_t dstep = out_k32_rows ? (int64_t)out_k32_rows * 32 : 32;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int q0 = d * 16 + (2 * kk) * 4, q1 = q0 + 4;
        uint32_t a0 = pack_bf16x2(o[q0 + 0] * inv, o[q0 + 1] * inv), a1 = pack_bf16x2(o[q0 + 2] * inv, o[q0 + 3] * inv);
        uint32_t b0 = pack_bf16x2(o[q1 + 0] * inv, o[q1 + 1] * inv), b1 = pack_bf16x2(o[q1 + 2] * inv, o[q1 + 3] * inv);
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                     : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
        // lanes 0-31: a = [own cols 16kk..+3 | upper half's cols 16kk+4..+7]; lanes 32-63: b = [lower's 16kk+8.. | own 16kk+12..]
        const u32x4_t w = {a0, a1, b0, b1};
        if (qrow < seq_len) *reinterpret_cast<u32x4_t*>(op + d * dstep + kk * 16) = w;
      }
  }
}

// Merge of the split-mode partials: out[row] = sum_s 2^(m_s - M) O_s[row] / sum_s 2^(m_s - M) l_s, M = max_s m_s.  One workgroup per
// (item, head), a thread = 32 channels (one 64-byte piece of the output row, row-major or K32-blocked) of one of the <= 64 rows.
__global__ __launch_bounds__(256) void attn_split_combine_kernel(uint16_t* __restrict__ out, int64_t ldo, int out_k32_rows,
                                                                 const int32_t* __restrict__ cu_seqlens, int H, int qfull, int nsplit,
                                                                 const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                                 const int32_t* __restrict__ item_skip) {
  const int hb = blockIdx.x, b = hb / H, h = hb - b * H;
  if (item_skip && item_skip[b]) return;
  const int seq_start = cu_seqlens[b], seq_len = cu_seqlens[b + 1] - seq_start;
  const int row = threadIdx.x >> 2, cg = threadIdx.x & 3;
  const int qrow = qfull * QBLK + row;
  if (qrow >= seq_len) return;
  float M = -INFINITY;
  for (int s = 0; s < nsplit; ++s) {
    const float* ml = part_ml + ((int64_t)(hb * nsplit + s) * 64 + row) * 2;
    if (ml[1] > 0.0f) M = fmaxf(M, ml[0]);
  }
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
  float L = 0.0f;
  for (int s = 0; s < nsplit; ++s) {
    const int64_t prow = (int64_t)(hb * nsplit + s) * 64 + row;
    const float l = part_ml[prow * 2 + 1];
    if (!(l > 0.0f)) continue;                      // an empty range
    const float w = __builtin_amdgcn_exp2f(part_ml[prow * 2] - M);
    L += w * l;
    const f32x4_t* po = reinterpret_cast<const f32x4_t*>(part_o + prow * DH + cg * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4_t x = po[i];
      acc[4 * i + 0] += w * x[0]; acc[4 * i + 1] += w * x[1]; acc[4 * i + 2] += w * x[2]; acc[4 * i + 3] += w * x[3];
    }
  }
  const float inv = 1.0f / L;
  uint16_t* op = out_k32_rows ? out + ((int64_t)(h * 4 + cg) * out_k32_rows + seq_start + qrow) * 32
                              : out + (int64_t)(seq_start + qrow) * ldo + h * DH + cg * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u32x4_t w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack_bf16x2(acc[8 * i + 2 * e] * inv, acc[8 * i + 2 * e + 1] * inv);
    *reinterpret_cast<u32x4_t*>(op + 8 * i) = w;
  }
}

// Split factor of the short last q-block (1 = off) from a makespan model in units of one full workgroup: `nfullwg` full blocks
// on `cus` CUs, then nh * ns pieces of 1 / ns that the dispatcher hands to whichever CU frees up.
int w64_split_factor(int nh, int qfull, int max_seqlen, int cus, size_t ws_bytes) {
  static const int knob = omni_dev_env_int("OMNI_ATTN_SPLIT", 1);        // dev knob (-DOMNI_DEV builds only)
  const int rem = max_seqlen % QBLK;
  if (!knob || rem == 0 || rem > 64 || qfull < 4) return 1;
  const long nfullwg = (long)nh * qfull, fr = nfullwg / cus, r = nfullwg % cus;
  auto makespan = [&](int ns) {
    const long small = (long)nh * ns;
    if (r > 0) {
      const long cap = (cus - r) * ns;
      return small <= cap ? (double)(fr + 1) : fr + 1 + (double)((small - cap + cus - 1) / cus) / ns;
    }
    return fr + (double)((small + cus - 1) / cus) / ns;
  };
  const int ntiles = (max_seqlen + KVBLK - 1) / KVBLK;
  int best = 1;
  double tbest = makespan(1) - 0.12;                // a split must save more than the partial round trip + the combine launch cost
  for (int ns = 2; ns <= 8; ++ns) {
    if (ntiles / ns < 8 || (size_t)nh * ns * 64 * (DH + 2) * sizeof(float) > ws_bytes) break;
    const double t = makespan(ns);
    if (t < tbest - 1e-9) { tbest = t; best = ns; }
  }
  return best;
}

}  // namespace

size_t omni_internal_flash_attn_w64_ws_bytes(int32_t B, int32_t H) {
  // room for the largest split the factor rule can pick (8 ranges) — 33 KB per (item, head) and range
  return (size_t)B * H * 8 * 64 * (DH + 2) * sizeof(float);
}

// internal: the 64-queries-per-wave kernel (same contract as omni_internal_flash_attn; picked by it for large grids).
// `part_ws` (nullable, fp32-aligned DEVICE memory of part_ws_bytes): enables split mode for the short last q-block.
int omni_internal_flash_attn_w64(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out, int64_t ldq,
                                 int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* cu_seqlens, int32_t B, int32_t H,
                                 int32_t max_seqlen, float softmax_scale, int32_t out_k32_rows, const int32_t* item_skip,
                                 int32_t q_prescaled, void* part_ws, size_t part_ws_bytes, void* stream) {
  static std::atomic<uint64_t> attr_done{0};
  OMNI_TRY_STATUS(omni_once_per_device(attr_done, [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_fwd_w64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               LDS_BYTES) == hipSuccess;
  }));
  const int qblocks = (max_seqlen + QBLK - 1) / QBLK;
  const int nh = B * H;
  int nsplit = 1;
  if (part_ws && (reinterpret_cast<uintptr_t>(part_ws) & 15) == 0)
    nsplit = w64_split_factor(nh, max_seqlen / QBLK, max_seqlen, omni_num_cus(), part_ws_bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (nsplit > 1) {
    const int qfull = max_seqlen / QBLK;
    float* part_o = static_cast<float*>(part_ws);
    float* part_ml = part_o + (size_t)nh * nsplit * 64 * DH;
    hipLaunchKernelGGL(flash_attn_fwd_w64_kernel, dim3(nh * qfull + nh * nsplit), dim3(256), LDS_BYTES, s, q, k, v, out, ldq, ldk,
                       ldv, ldo, cu_seqlens, nh, H, softmax_scale * 1.4426950408889634f, out_k32_rows, item_skip, q_prescaled, qfull,
                       nsplit, part_o, part_ml);
    OMNI_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_split_combine_kernel, dim3(nh), dim3(256), 0, s, out, ldo, out_k32_rows, cu_seqlens, H, qfull, nsplit,
                       part_o, part_ml, item_skip);
    OMNI_CHECK_LAUNCH();
    return OMNI_OK;
  }
  hipLaunchKernelGGL(flash_attn_fwd_w64_kernel, dim3(nh * qblocks), dim3(256), LDS_BYTES, s, q, k, v, out, ldq, ldk, ldv, ldo,
                     cu_seqlens, nh, H, softmax_scale * 1.4426950408889634f, out_k32_rows, item_skip, q_prescaled, qblocks, 1,
                     nullptr, nullptr);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}
