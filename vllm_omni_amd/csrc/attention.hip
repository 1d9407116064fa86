// Flash attention forward for gfx950: non-causal, no mask, head_dim 128, bf16 in/out, fp32 softmax.
//
// Production kernel: flash_attn_fwd_pipe_kernel<NW, NQ> (second half of this file; software-pipelined, LDS-DMA staging).
// flash_attn_fwd_kernel<NQ> (first half) is the unskewed, register-staged predecessor, kept for A/B runs
// (OMNI_ATTN_PIPE=0).  Both share the scheme below.
//
// One workgroup = NW waves x 32 (x NQ) query rows of one (item, head).  K/V tiles of 64 keys are staged through LDS
// (2 stages x (16 KiB K + 16 KiB V) = 64 KiB -> 2 workgroups / CU at 4 waves).
//
//   Sᵀ = K·Qᵀ  (SWAPPED operands): A-operand = K fragment from LDS, B-operand = Q fragment held in
//        registers for the whole kernel.  The 32x32 accumulator then puts ONE query in each lane
//        (col = lane&31) and 16 of the 32 keys of a sub-block in its registers, so the softmax max / sum /
//        rescale are lane-local; the two half-waves exchange one value per tile for the max.
//   Oᵀ = Vᵀ·Pᵀ: B-operand = Pᵀ is the Sᵀ accumulator itself (exp'ed, packed to bf16; the C-layout key order
//        4*hi + {0..3} (+8) is simply adopted as the MFMA k order), A-operand = Vᵀ fragment read from a
//        row-major V tile with ds_read_b64_tr_b16 (hardware 4x4 transpose).  Oᵀ keeps q = lane&31 per lane,
//        so the online-softmax rescale of O is a per-lane scalar multiply.
//   LDS images: K row-major [64][128] with the 16-B chunk index XOR (key&15) (conflict-free ds_read_b128);
//        V as [dblk 4][key/4 16][4 keys][32 d] (each 32-lane half of a tr-read covers one contiguous 256 B).
//
// Layout contract: q/k/v/out are [total_rows, H*128] (the reference's [B,S,H,dh] flattened), item b owns rows
// [cu_seqlens[b], cu_seqlens[b+1]).  Roofline: MFMA-bound, 4*S^2*128 flop per (item, head).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int DH = 128;
constexpr int NWAVES = 4;
constexpr int KVBLK = 64;
constexpr int K_TILE_BYTES = KVBLK * DH * 2;  // 16 KiB
constexpr int STAGE_BYTES = 2 * K_TILE_BYTES; // K + V
constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // 64 KiB

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;

template <int OFF>
OMNI_DEVINL bf16x8_t lds_read16(uint32_t addr) {   // invisible to hipcc's waitcnt pass: waits are counted by hand
  bf16x8_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}

template <int OFF>
OMNI_DEVINL u32x2_t lds_tr_read8(uint32_t addr) {   // ds_read_b64_tr_b16: hardware 4x4 transpose read
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}

OMNI_DEVINL bf16x8_t tr_read_pair(uint32_t lds_addr_a, uint32_t lds_addr_b) {
  // two hardware-transposed 4x(16 lanes) reads -> the 8 k-elements of one MFMA A fragment
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)lds_addr_a);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)lds_addr_b);
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

#ifdef OMNI_DEV
#include "dev/attention_family_round1_register_staged.inc"
#endif

// ------------------------------------------------------------------------------------------------
// Software-pipelined variant (default).  Same tiling and LDS images as flash_attn_fwd_kernel<1>, but
//   * K/V tiles are staged by LDS-DMA (global_load_lds_dwordx4; the XOR swizzle / the V block layout are applied to the
//     per-lane SOURCE address) instead of global -> 32 staging VGPRs -> ds_write: the registers pay for a second S tile;
//   * the loop is skewed by one tile: iteration t runs  QK^T(t+1) || exp-half of softmax(t)  and then
//     P.V(t) || max-half of softmax(t+1).  An MFMA occupies the matrix pipe for 32 cycles but its wave only for the
//     issue, so the VALU work of the *same wave* placed between two MFMAs runs under them; in the unskewed loop a wave's
//     softmax can only be hidden by the OTHER wave of the SIMD happening to be in an MFMA phase (measured: matrix pipe
//     busy ~40 %, with VALU ~= MFMA cycles per tile).
// Tile j lives in LDS stage j&1.  K(t+2) is fetched into stage t&1 at the top of iteration t (K(t) was consumed one
// iteration earlier), V(t+1) into stage (t+1)&1 (V(t-1) was consumed at the end of iteration t-1); both land before the
// single barrier that ends the iteration.
// ------------------------------------------------------------------------------------------------
// LDS-DMA in the SADDR form (uniform 64-bit tile base in SGPRs + one 32-bit per-lane byte offset): the builtin keeps a
// zero-extended 64-bit VGPR pair per source (16 registers for the 8 sources, and 64-bit VALU adds per tile).  Issued from
// inline asm, so the loads are invisible to hipcc's waitcnt pass: the loop retires them with its own vmcnt(0).  M0 (the
// wave-uniform LDS byte address; the hardware adds lane*16) is written here and used by nothing else in the kernel.
OMNI_DEVINL void attn_glds16(const char* tile_base, uint32_t lane_byte_off, uint32_t lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds_byte_addr), "v"(lane_byte_off), "s"(tile_base) : "memory");
}
// max over the two half-waves WITHOUT touching LDS (a ds_bpermute would break the counted lgkmcnt waits of the
// fragment pipeline): v_permlane32_swap exchanges a[32..63] with b[0..31].  Inline asm with two distinct registers —
// the builtin called with the same value for both operands is folded to one register by hipcc.
OMNI_DEVINL float xhalf_max(float x) {
  uint32_t a = __builtin_bit_cast(uint32_t, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
OMNI_DEVINL float xhalf_sum(float x) {
  uint32_t a = __builtin_bit_cast(uint32_t, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

#ifndef OMNI_ATTN_DMA_BURST
#define OMNI_ATTN_DMA_BURST 1   // 1: a tile's DMA pieces in one burst at the top of the iteration (measured 881 / 913 TF/s
                                // at 4 / 8 waves); 0: one piece per P.V MFMA slot (840 / 865: the issue stalls land on the
                                // MFMA-bound phase)
#endif
#ifndef OMNI_ATTN_SETPRIO
#define OMNI_ATTN_SETPRIO 1
#endif
#ifndef OMNI_ATTN_PKFMA
#define OMNI_ATTN_PKFMA 0   // 1: v_pk_fma_f32 from inline asm for the exponent argument (measured -3 %: pair set-up moves)
#endif
#ifndef OMNI_ATTN_BAKE
#define OMNI_ATTN_BAKE 1  // 1: Q is pre-scaled by softmax_scale*log2(e) (by the QKV GEMM epilogue, or in this kernel's prologue)
                          // and the first MFMA of every S chain takes C = -m_run (a 16-register splat, rewritten only on the
                          // rare rescale path) instead of 0: the accumulator then IS the exp2 argument s~ - m~, and the 32
                          // v_fma per wave and tile that formed it are gone (the loop is bound by per-wave instruction issue)
#endif
#ifndef OMNI_ATTN_W64
#define OMNI_ATTN_W64 1   // large grids run flash_attn_fwd_w64_kernel (attention_w64.hip)
#endif
#ifndef OMNI_ATTN_ABL
#define OMNI_ATTN_ABL 0   // dev-only timing ablations (wrong results): 1 no DMA wait, 2 no barrier, 4 no DMA, 8 no exp, 16 no LDS reads
#endif
// NW = waves per workgroup sharing one K/V tile stream: 4 (128 queries, 2 workgroups per CU) or 8 (256 queries, one
// workgroup per CU: half the LDS-DMA write traffic per query; the LDS write port is shared with the fragment reads).
// NQ = 32-query blocks per wave.  NQ = 2 (with NW = 4: 256 queries per workgroup, ONE wave per SIMD and its whole
// 512-entry register file): every K / V^T fragment read from LDS feeds two MFMAs and the DMA per query halves — the two
// resources the ablations show this loop to be limited by.  The accumulators then live in AGPRs (an inline
// asm with an "a" constraint switches hipcc to the AGPR form of the MFMAs; without it the second half of the register file
// is only used as spill space).
// PP = 1 (8 waves, NQ = 1): the two waves of every SIMD run HALF AN ITERATION APART.  An iteration has a VALU-heavy half
// (H1: QK^T(t+1) MFMAs with the 32 exp / 32 fma / pack of softmax(t) between them: ~52 VALU cycles per 32-cycle MFMA) and an
// MFMA-bound half (H2: P.V(t), two v_max3 per MFMA).  With all 8 waves in the same half (PP = 0) a SIMD's two waves ask for
// ~104 VALU cycles per 64 matrix-pipe cycles in H1 and leave the VALU idle in H2: the matrix pipe is busy 49 % (measured).
// With wave group B (waves 4..7, the SIMD partners of waves 0..3) one half behind, every half-phase pairs one wave's H1
// with its partner's H2.  A barrier ends every half-phase; DMA pieces are issued by wall-clock half-phase h (K(h/2 + 2) at the
// start of even h, V((h+1)/2) at odd h, by all waves: group A from H1 / H2 of its iteration t = h/2, group B from H2 / H1),
// and retired with a counted vmcnt at the end of the NEXT half-phase:
//   RAW: K(t+1) is read in H1(t) (A: h = 2t, B: 2t+1), issued at h = 2t-2, waited (vmcnt <= the pieces of h = 2t-1) at the
//        end of h = 2t-1;  V(t) is read in H2(t) (h = 2t+1 / 2t+2), issued at h = 2t-1, waited at the end of h = 2t.
//   WAR: K(t+2) overwrites K(t) (last read by B at h = 2t-1) at h = 2t;  V(t+1) overwrites V(t-1) (last read by B at h = 2t)
//        at h = 2t+1.
template <int NW, int NQ = 1, int PP = 0>
__global__ __launch_bounds__(NW * 64, ((NW == 4 && NQ == 1) ? 2 : 1)) void flash_attn_fwd_pipe_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    uint16_t* __restrict__ out, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
    const int32_t* __restrict__ cu_seqlens, int n_heads_total, int H, float scale_log2e, int out_k32_rows,
    int block_order, const int32_t* __restrict__ item_skip, int q_prescaled) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  // Block -> (item*head, q-block).  block_order 1 (default): XCD-aware.  Block b runs on XCD b % 8; each XCD gets a
  // CONTIGUOUS range of the (head-major, q-block-minor) work list, so the ~32 workgroups resident on an XCD are all the
  // q-blocks of about two heads: they stream the same 2 MB of K/V through that XCD's 4 MiB L2 at the same pace (one miss +
  // 16 hits per line).  block_order 0 (round 1): heads fastest — an XCD then holds 18 different heads at once (38 MB of
  // K/V against 4 MiB of L2: TCC hit 49 %, 5.4x the algorithmic fabric traffic, profiles/r01).
  int hb, qb;
  if (block_order == 1) {
    const int nwg = gridDim.x, bid = blockIdx.x, qblocks = nwg / n_heads_total;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    hb = lid / qblocks;
    qb = lid - hb * qblocks;
  } else {
    hb = blockIdx.x % n_heads_total;
    qb = blockIdx.x / n_heads_total;
  }
  const int b = hb / H, h = hb - b * H;
  if (item_skip && item_skip[b]) return;          // device-side predicate: this item's block stack is skipped (omni_teacache)
  const int seq_start = cu_seqlens[b];
  const int seq_len = cu_seqlens[b + 1] - seq_start;
  constexpr int QBLK = 32 * NW * NQ;
  constexpr int NPIECE = 16 / NW;   // DMA pieces (1 KiB) per wave per operand per tile
  static_assert(!PP || (NW == 8 && NQ == 1), "ping-pong: 8 waves (2 DMA pieces per wave and operand: the counted vmcnt(2))");
  if (qb * QBLK >= seq_len) return;

  const char* kbase = reinterpret_cast<const char*>(k + (int64_t)seq_start * ldk + h * DH);
  const char* vbase = reinterpret_cast<const char*>(v + (int64_t)seq_start * ldv + h * DH);

  if (NQ == 2) {
    float agpr_form_ = 0.f;
    asm volatile("" : "+a"(agpr_form_));   // see the note above the kernel
  }
  bf16x8_t qf[NQ][8];
#pragma unroll
  for (int bq = 0; bq < NQ; ++bq) {
    const int qrow = min(qb * QBLK + (wave * NQ + bq) * 32 + l31, seq_len - 1);
    const uint16_t* qp = q + (int64_t)(seq_start + qrow) * ldq + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[bq][ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
    if (OMNI_ATTN_BAKE && !q_prescaled) {
      // callers outside the fused DiT block hand over the reference's un-scaled q: Q~ = bf16(q * scale * log2 e) here (one
      // extra bf16 rounding of q; the fused QKV epilogue folds the factor into its fp32 RMSNorm weight instead)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        u32x4_t w = __builtin_bit_cast(u32x4_t, qf[bq][ks]);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf16x2(bf16_lo(w[e]) * scale_log2e, bf16_hi(w[e]) * scale_log2e);
        qf[bq][ks] = __builtin_bit_cast(bf16x8_t, w);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[bq][ks]));   // see flash_attn_fwd_kernel: no rematerialisation
  }

  // ---- DMA sources.  One wave-instruction = 1 KiB of the LDS image, lane L -> byte 16*L of the piece.
  //  K piece p = wave + 4i (keys 4p .. 4p+3):   key = 4p + (L>>4), LDS chunk L&15 holds logical chunk (L&15)^(key&15)
  //  V piece (dblk = i, key group = wave):       key = 16*wave + 4*(L>>4) + ((L>>2)&3), logical chunk 4i + (L&3)
  //    (the four dblk pieces of one key group are issued back to back: together they cover whole 256-B V rows)
  // Only the 8 per-lane byte offsets stay live across the loop; the ragged-tail variant recomputes key / column.
  // piece P = wave + NW*i of each operand image (16 pieces of 1 KiB); V piece P = (dblk = P>>2, key group = P&3)
  auto k_key_of = [&](int i) { return 4 * (wave + NW * i) + (lane >> 4); };
  auto k_col_of = [&](int i) { return (uint32_t)(((lane & 15) ^ (k_key_of(i) & 15)) * 16); };
  auto v_key_of = [&](int i) { return 16 * ((wave + NW * i) & 3) + 4 * (lane >> 4) + ((lane >> 2) & 3); };
  auto v_col_of = [&](int i) { return (uint32_t)((4 * ((wave + NW * i) >> 2) + (lane & 3)) * 16); };
  uint32_t k_src[NPIECE], v_src[NPIECE];
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) {
    k_src[i] = (uint32_t)(k_key_of(i) * ldk * 2) + k_col_of(i);
    v_src[i] = (uint32_t)(v_key_of(i) * ldv * 2) + v_col_of(i);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int ntiles = (seq_len + KVBLK - 1) / KVBLK;
  const int last_valid = seq_len - (ntiles - 1) * KVBLK;          // keys in the last tile (1..64)
  // one DMA piece (a global_load_lds costs its wave 60-185 issue cycles: they are spread over the P.V MFMAs, not burst)
  auto issue_K_piece = [&](int t, int stage, int i) {
    const char* tb = kbase + (int64_t)t * KVBLK * ldk * 2;          // uniform
    const uint32_t dst = lds0 + stage * STAGE_BYTES + (wave + NW * i) * 1024;
    if (t == ntiles - 1 && last_valid < KVBLK)                      // ragged tail: re-read the last valid row
      attn_glds16(tb, (uint32_t)(min(k_key_of(i), last_valid - 1) * ldk * 2) + k_col_of(i), dst);
    else
      attn_glds16(tb, k_src[i], dst);
  };
  auto issue_V_piece = [&](int t, int stage, int i) {
    const char* tb = vbase + (int64_t)t * KVBLK * ldv * 2;
    const uint32_t dst = lds0 + stage * STAGE_BYTES + K_TILE_BYTES + (wave + NW * i) * 1024;
    if (t == ntiles - 1 && last_valid < KVBLK)
      attn_glds16(tb, (uint32_t)(min(v_key_of(i), last_valid - 1) * ldv * 2) + v_col_of(i), dst);
    else
      attn_glds16(tb, v_src[i], dst);
  };
  auto issue_K = [&](int t, int stage) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) issue_K_piece(t, stage, i);
  };
  auto issue_V = [&](int t, int stage) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) issue_V_piece(t, stage, i);
  };

  uint32_t k_addr[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_addr[ks] = lds0 + l31 * 256 + ((((uint32_t)(ks * 2 + hi)) ^ (l31 & 15)) << 4);
  // absolute LDS addresses; the stage (a compile-time constant per call site: the loop is unrolled by two) and the fragment
  // index go into the 16-bit offset immediate of the ds_reads, so the loop has no address arithmetic at all
  const uint32_t v_addr = lds0 + K_TILE_BYTES + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8 + hi * 256;

  f32x16_t o[NQ][4];
  float m_run[NQ], l_run[NQ];
  f32x16_t negm16[NQ];       // OMNI_ATTN_BAKE: -m_run (exp2 domain) in all 16 registers = the C operand that opens an S chain
#pragma unroll
  for (int bq = 0; bq < NQ; ++bq) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[bq][d][i] = 0.0f;
    m_run[bq] = OMNI_ATTN_BAKE ? 0.0f : -INFINITY;     // BAKE: m_run holds -m~ (what negm16 is a splat of); tile 0 always rescales
    l_run[bq] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm16[bq][i] = 0.0f;
  }

  bf16x8_t kf[4];
#define OMNI_KREAD(i, kst)   /* kst = LDS stage (compile-time 0 / 1) */                                              \
  do {                                                                                                              \
    if (OMNI_ATTN_ABL & 16) asm volatile("" : "=v"(kf[(i) & 3]) : "v"(k_addr[(i) & 7]));                             \
    else kf[(i) & 3] = lds_read16<((i) >> 3) * 32 * 256 + (kst) * STAGE_BYTES>(k_addr[(i) & 7]);                    \
  } while (0)
#define OMNI_QK_STEP(i, SN, kst, CHUNK)                                                          \
  do {                                                                                           \
    if ((i) <= 12) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");                            \
    else if ((i) == 13) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                       \
    else if ((i) == 14) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");                       \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    _Pragma("unroll") for (int bq_ = 0; bq_ < NQ; ++bq_)                                          \
      SN[bq_][(i) >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[(i) & 3], qf[bq_][(i) & 7],               \
                              (OMNI_ATTN_BAKE && ((i) & 7) == 0) ? negm16[bq_] : SN[bq_][(i) >> 3], 0, 0, 0);          \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if ((i) + 4 < 16) OMNI_KREAD((i) + 4, kst);                                                  \
    CHUNK;                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
#define OMNI_QK_ALL(SN, kst, CH)                                                                                     \
  do {                                                                                                               \
    OMNI_KREAD(0, kst); OMNI_KREAD(1, kst); OMNI_KREAD(2, kst); OMNI_KREAD(3, kst);                                  \
    OMNI_QK_STEP(0, SN, kst, CH(0));   OMNI_QK_STEP(1, SN, kst, CH(1));   OMNI_QK_STEP(2, SN, kst, CH(2));           \
    OMNI_QK_STEP(3, SN, kst, CH(3));   OMNI_QK_STEP(4, SN, kst, CH(4));   OMNI_QK_STEP(5, SN, kst, CH(5));           \
    OMNI_QK_STEP(6, SN, kst, CH(6));   OMNI_QK_STEP(7, SN, kst, CH(7));   OMNI_QK_STEP(8, SN, kst, CH(8));           \
    OMNI_QK_STEP(9, SN, kst, CH(9));   OMNI_QK_STEP(10, SN, kst, CH(10)); OMNI_QK_STEP(11, SN, kst, CH(11));         \
    OMNI_QK_STEP(12, SN, kst, CH(12)); OMNI_QK_STEP(13, SN, kst, CH(13)); OMNI_QK_STEP(14, SN, kst, CH(14));         \
    OMNI_QK_STEP(15, SN, kst, CH(15));                                                                               \
  } while (0)
#define OMNI_NOCHUNK(i) (void)0

  auto mask_tail = [&](f32x16_t (&S)[NQ][2], int kv0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kv0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key >= seq_len) {
#pragma unroll
          for (int bq = 0; bq < NQ; ++bq) S[bq][j][r] = -INFINITY;
        }
      }
  };
  auto zero_s = [&](f32x16_t (&S)[NQ][2]) {
#pragma unroll
    for (int bq = 0; bq < NQ; ++bq)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) S[bq][j][i] = 0.0f;
  };

  // ---- prologue: tiles 0 (K, V) and 1 (K) in flight, S(0) and its row max -------------------------------------
  issue_K(0, 0);
  issue_V(0, 0);
  if (ntiles > 1) issue_K(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16_t sA[NQ][2], sB[NQ][2];
  float mxA[NQ], mxB[NQ];
  if (!OMNI_ATTN_BAKE) zero_s(sA);
  OMNI_QK_ALL(sA, 0, OMNI_NOCHUNK);
  if (KVBLK > seq_len) mask_tail(sA, 0);
#pragma unroll
  for (int bq = 0; bq < NQ; ++bq) {
    float mx = sA[bq][0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sA[bq][0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sA[bq][1][r]);
    mxA[bq] = xhalf_max(mx);
    mxB[bq] = 0.0f;
  }
  const bool grpB = PP && wave >= NW / 2;
  if (PP) {
    __syncthreads();                               // every wave has read K(0): its stage can take K(2)
    if (ntiles > 2) issue_K(2, 0);
    if (grpB) __builtin_amdgcn_s_barrier();        // group B sits out half-phase 0
  }

  // One iteration; SC = S(t) (complete, row max mxc known), SN receives S(t+1).
  // has_next is a compile-time constant: a run-time flag puts a branch between every two MFMAs of the P.V phase.
  // par_c = t & 1 as a compile-time constant (tile j lives in stage j & 1)
  auto iteration = [&](auto has_next_c, auto par_c, int t, f32x16_t (&SC)[NQ][2], f32x16_t (&SN)[NQ][2], float (&mxc)[NQ],
                       float (&mxn)[NQ]) {
    constexpr bool has_next = decltype(has_next_c)::value;
    constexpr int PAR = decltype(par_c)::value;
    const bool dma_k = !(OMNI_ATTN_ABL & 4) && t + 2 < ntiles, dma_v = !(OMNI_ATTN_ABL & 4) && has_next;
    bool issued1 = false, issued2 = false;         // PP: pieces issued at the start of this wave's H1 / H2
    if (PP) {
      if (!grpB) { if (t >= 1 && dma_k) { issue_K(t + 2, t & 1); issued1 = true; } }
      else if (dma_v) { issue_V(t + 1, (t + 1) & 1); issued1 = true; }
    } else if (OMNI_ATTN_DMA_BURST == 1) {
      if (dma_k) issue_K(t + 2, t & 1);
      if (dma_v) issue_V(t + 1, (t + 1) & 1);
    }

    // defer-max decision (see flash_attn_fwd_kernel); the rescale is rare after the first tiles
    constexpr float DEFER = 6.0f;
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t scale2 = {scale_log2e, scale_log2e};
    f32x2_t mneg2[NQ], psum2[NQ];
    uint32_t pfu[NQ][2][2][4];
#pragma unroll
    for (int bq = 0; bq < NQ; ++bq) {
      if (OMNI_ATTN_BAKE) {
        // SC already holds s~ - m~ (m~ = the running max when its chain was opened) and mxc is its row max: the rescale is
        // needed only when that exceeds the defer threshold.  Tile 0 opened its chain with C = 0: it always "rescales"
        // (o = l = 0), which sets the first max — also when every score of the tile is far below zero.
        if (t == 0 || !__all(mxc[bq] <= DEFER)) {
          const float d = t == 0 ? mxc[bq] : fmaxf(mxc[bq], 0.0f);
          // t == 0: o = l = 0, nothing to rescale — and exp2(-d) would be +inf when every score of the first tile sits below
          // about -128 in the exp2 domain (0 * inf = NaN for the whole output row): only the shift by d applies there
          const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
          l_run[bq] *= alpha;
#pragma unroll
          for (int dd = 0; dd < 4; ++dd)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[bq][dd][i] *= alpha;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) SC[bq][j][i] -= d;
          m_run[bq] -= d;
#pragma unroll
          for (int i = 0; i < 16; ++i) negm16[bq][i] = m_run[bq];
        }
      } else {
      if (!__all((mxc[bq] - m_run[bq]) * scale_log2e <= DEFER)) {
        const float m_new = fmaxf(m_run[bq], mxc[bq]);
        const float alpha = __builtin_amdgcn_exp2f((m_run[bq] - m_new) * scale_log2e);
        m_run[bq] = m_new;
        l_run[bq] *= alpha;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[bq][d][i] *= alpha;
      }
      const float mneg = -m_run[bq] * scale_log2e;
      mneg2[bq][0] = mneg; mneg2[bq][1] = mneg;
      }
      psum2[bq][0] = 0.0f; psum2[bq][1] = 0.0f;
    }
    // exp chunk i: S elements (flat index over j, r) 2i and 2i+1 -> one packed bf16 pair of P.  The sum / pack of chunk i
    // is issued in chunk i+1, behind that chunk's exps: a v_exp result needs a wait state before a VALU may read it, and
    // with the consumer right behind hipcc fills it with an s_nop (one issue slot per chunk)
    f32x2_t pend[NQ];
#define OMNI_EXP_FINISH(i)                                                                                       \
  do {                                                                                                           \
    _Pragma("unroll") for (int bq_ = 0; bq_ < NQ; ++bq_) {                                                        \
      psum2[bq_] += pend[bq_];                                                                                   \
      uint32_t pk_ = pack_bf16x2(pend[bq_][0], pend[bq_][1]);                                                    \
      asm volatile("" : "+v"(pk_), "+v"(psum2[bq_])); /* pin: pure arithmetic is otherwise sunk below the MFMA run */ \
      pfu[bq_][(i) >> 3][((i) >> 2) & 1][(i) & 3] = pk_;                                                         \
    }                                                                                                            \
  } while (0)
#define OMNI_EXP_CHUNK(i)                                                                                        \
  do {                                                                                                           \
    if (OMNI_ATTN_ABL & 8) {                                                                                     \
      _Pragma("unroll") for (int bq_ = 0; bq_ < NQ; ++bq_) pfu[bq_][(i) >> 3][((i) >> 2) & 1][(i) & 3] = 0x3c003c00u; \
      break;                                                                                                     \
    }                                                                                                            \
    f32x2_t e_[NQ];                                                                                              \
    _Pragma("unroll") for (int bq_ = 0; bq_ < NQ; ++bq_) {                                                        \
      /* packed fp32: S elements 2i, 2i+1 are an aligned register pair */                                        \
      const f32x2_t x_ = {SC[bq_][(i) >> 3][(2 * (i)) & 15], SC[bq_][(i) >> 3][((2 * (i)) & 15) + 1]};             \
      f32x2_t y_;                                                                                                \
      if (OMNI_ATTN_BAKE) y_ = x_;                                                                               \
      else if (OMNI_ATTN_PKFMA) asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(y_) : "v"(x_), "v"(scale2), "v"(mneg2[bq_])); \
      else y_ = __builtin_elementwise_fma(x_, scale2, mneg2[bq_]);                                               \
      e_[bq_][0] = __builtin_amdgcn_exp2f(y_[0]);                                                                \
      e_[bq_][1] = __builtin_amdgcn_exp2f(y_[1]);                                                                \
      asm volatile("" : "+v"(e_[bq_]));                                                                          \
    }                                                                                                            \
    if ((i) > 0) OMNI_EXP_FINISH(((i) + 15) & 15);                                                               \
    _Pragma("unroll") for (int bq_ = 0; bq_ < NQ; ++bq_) pend[bq_] = e_[bq_];                                     \
  } while (0)
    if (has_next) {
      if (!OMNI_ATTN_BAKE) zero_s(SN);
      if (OMNI_ATTN_SETPRIO) __builtin_amdgcn_s_setprio(1);
      OMNI_QK_ALL(SN, (PAR ^ 1), OMNI_EXP_CHUNK);      // K(t+1) lives in stage (t+1) & 1
      if (OMNI_ATTN_SETPRIO) __builtin_amdgcn_s_setprio(0);
    } else {
      OMNI_EXP_CHUNK(0);  OMNI_EXP_CHUNK(1);  OMNI_EXP_CHUNK(2);  OMNI_EXP_CHUNK(3);  OMNI_EXP_CHUNK(4);  OMNI_EXP_CHUNK(5);
      OMNI_EXP_CHUNK(6);  OMNI_EXP_CHUNK(7);  OMNI_EXP_CHUNK(8);  OMNI_EXP_CHUNK(9);  OMNI_EXP_CHUNK(10); OMNI_EXP_CHUNK(11);
      OMNI_EXP_CHUNK(12); OMNI_EXP_CHUNK(13); OMNI_EXP_CHUNK(14); OMNI_EXP_CHUNK(15);
    }
    if (!(OMNI_ATTN_ABL & 8)) OMNI_EXP_FINISH(15);
#undef OMNI_EXP_CHUNK
#undef OMNI_EXP_FINISH
#pragma unroll
    for (int bq = 0; bq < NQ; ++bq) l_run[bq] += psum2[bq][0] + psum2[bq][1];
    if (has_next && (t + 2) * KVBLK > seq_len) mask_tail(SN, (t + 1) * KVBLK);

    if (PP) {                                      // ---- end of H1: the pieces of the previous half-phase have landed
      if (issued1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (!grpB) { if (dma_v) { issue_V(t + 1, (t + 1) & 1); issued2 = true; } }
      else if (!(OMNI_ATTN_ABL & 4) && t + 3 < ntiles) { issue_K(t + 3, (t + 1) & 1); issued2 = true; }
    }

    // ---- O^T += V^T P^T (tile t), with the row max of S(t+1) in the MFMA shadows
    float mx[NQ];
#pragma unroll
    for (int bq = 0; bq < NQ; ++bq) mx[bq] = has_next ? SN[bq][0][0] : 0.0f;
    {
      const uint32_t vb = v_addr;                 // V(t) lives in stage t & 1 = PAR: folded into the offset immediates
      u32x2_t vlo[4], vhi[4];
#define OMNI_VOFF(i) (PAR * STAGE_BYTES + ((i) & 3) * 4096 + ((((i) >> 3) * 8 + (((i) >> 2) & 1) * 4) * 256))
#define OMNI_VREAD(i)                                                                      \
  do {                                                                                     \
    if (OMNI_ATTN_ABL & 16) {                                                              \
      asm volatile("" : "=v"(vlo[(i) & 3]) : "v"(vb));                                     \
      asm volatile("" : "=v"(vhi[(i) & 3]) : "v"(vb));                                     \
    } else {                                                                               \
      vlo[(i) & 3] = lds_tr_read8<OMNI_VOFF(i)>(vb);                                       \
      vhi[(i) & 3] = lds_tr_read8<OMNI_VOFF(i) + 512>(vb);                                 \
    }                                                                                      \
  } while (0)
#define OMNI_PV(i)                                                                                             \
  do {                                                                                                         \
    if ((i) <= 13) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                          \
    else if ((i) == 14) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                                     \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    {                                                                                                          \
      const u32x4_t w_ = {vlo[(i) & 3][0], vlo[(i) & 3][1], vhi[(i) & 3][0], vhi[(i) & 3][1]};                 \
      _Pragma("unroll") for (int bq_ = 0; bq_ < NQ; ++bq_) {                                                    \
        const u32x4_t p_ = {pfu[bq_][(i) >> 3][((i) >> 2) & 1][0], pfu[bq_][(i) >> 3][((i) >> 2) & 1][1],      \
                            pfu[bq_][(i) >> 3][((i) >> 2) & 1][2], pfu[bq_][(i) >> 3][((i) >> 2) & 1][3]};     \
        o[bq_][(i) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w_),            \
                                                    __builtin_bit_cast(bf16x8_t, p_), o[bq_][(i) & 3], 0, 0, 0); \
      }                                                                                                        \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#define OMNI_MAX_CHUNK(c)                                                                                       \
  do {                                                                                                          \
    if (has_next) {                                                                                             \
      _Pragma("unroll") for (int bq_ = 0; bq_ < NQ; ++bq_) {                                                     \
        mx[bq_] = fmaxf(fmaxf(mx[bq_], SN[bq_][(c) >> 2][((c) & 3) * 4 + 0]), SN[bq_][(c) >> 2][((c) & 3) * 4 + 1]); \
        mx[bq_] = fmaxf(fmaxf(mx[bq_], SN[bq_][(c) >> 2][((c) & 3) * 4 + 2]), SN[bq_][(c) >> 2][((c) & 3) * 4 + 3]); \
        asm volatile("" : "+v"(mx[bq_]));                                                                       \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)
// DMA slot j (0 .. 2*NPIECE-1): K(t+2) pieces first (needed one iteration from now), then V(t+1) pieces
#define OMNI_DMA_SLOT(j)                                                                  \
  do {                                                                                    \
    if (!OMNI_ATTN_DMA_BURST && (j) < 2 * NPIECE) {                                       \
      if ((j) < NPIECE) { if (dma_k) issue_K_piece(t + 2, t & 1, (j) % NPIECE); }         \
      else { if (dma_v) issue_V_piece(t + 1, (t + 1) & 1, (j) % NPIECE); }                \
      __builtin_amdgcn_sched_barrier(0);                                                  \
    }                                                                                     \
  } while (0)
      OMNI_VREAD(0); OMNI_VREAD(1); OMNI_VREAD(2);
      if (OMNI_ATTN_SETPRIO) __builtin_amdgcn_s_setprio(1);
      OMNI_PV(0);  OMNI_VREAD(3);  OMNI_DMA_SLOT(0); OMNI_PV(1);  OMNI_VREAD(4);  OMNI_DMA_SLOT(1);
      OMNI_PV(2);  OMNI_VREAD(5);  OMNI_DMA_SLOT(2); OMNI_PV(3);  OMNI_VREAD(6);  OMNI_DMA_SLOT(3);
      OMNI_PV(4);  OMNI_VREAD(7);  OMNI_MAX_CHUNK(0); OMNI_PV(5);  OMNI_VREAD(8);  OMNI_MAX_CHUNK(1);
      OMNI_PV(6);  OMNI_VREAD(9);  OMNI_MAX_CHUNK(2); OMNI_PV(7);  OMNI_VREAD(10); OMNI_MAX_CHUNK(3);
      OMNI_PV(8);  OMNI_VREAD(11); OMNI_MAX_CHUNK(4); OMNI_PV(9);  OMNI_VREAD(12); OMNI_MAX_CHUNK(5);
      OMNI_PV(10); OMNI_VREAD(13); OMNI_MAX_CHUNK(6); OMNI_PV(11); OMNI_VREAD(14); OMNI_MAX_CHUNK(7);
      OMNI_PV(12); OMNI_VREAD(15); OMNI_DMA_SLOT(4); OMNI_PV(13); OMNI_DMA_SLOT(5); OMNI_PV(14); OMNI_DMA_SLOT(6);
      OMNI_PV(15); OMNI_DMA_SLOT(7);
      if (OMNI_ATTN_SETPRIO) __builtin_amdgcn_s_setprio(0);
#undef OMNI_DMA_SLOT
#undef OMNI_MAX_CHUNK
#undef OMNI_PV
#undef OMNI_VREAD
#undef OMNI_VOFF
    }
    if (has_next) {
#pragma unroll
      for (int bq = 0; bq < NQ; ++bq) mxn[bq] = xhalf_max(mx[bq]);
    }
    if (PP) {                                      // ---- end of H2
      if (issued2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      if (!(OMNI_ATTN_ABL & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this iteration's DMA has landed
      if (!(OMNI_ATTN_ABL & 2)) __syncthreads();          // ... for every wave; and every wave is done with K(t+1), V(t)
    }
  };

  {
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    using even = std::integral_constant<int, 0>;
    using odd = std::integral_constant<int, 1>;
    int t = 0;                                     // even at every call site below
    for (; t + 2 < ntiles; t += 2) {
      iteration(yes{}, even{}, t, sA, sB, mxA, mxB);
      iteration(yes{}, odd{}, t + 1, sB, sA, mxB, mxA);
    }
    if (t + 1 < ntiles) {
      iteration(yes{}, even{}, t, sA, sB, mxA, mxB);
      iteration(no{}, odd{}, t + 1, sB, sA, mxB, mxA);
    } else {
      iteration(no{}, even{}, t, sA, sB, mxA, mxB);
    }
  }
  if (PP && !grpB) __builtin_amdgcn_s_barrier();   // group A waits out group B's last half-phase
#undef OMNI_QK_ALL
#undef OMNI_QK_STEP
#undef OMNI_KREAD
#undef OMNI_NOCHUNK

  // ---- epilogue -------------------------------------------------------------------------------------------------
#pragma unroll
  for (int bq = 0; bq < NQ; ++bq) {
    const float inv = 1.0f / xhalf_sum(l_run[bq]);
    const int qrow = qb * QBLK + (wave * NQ + bq) * 32 + l31;
    if (qrow < seq_len) {
      // row-major: out[row][h*128 + d*32 + qd*8 + hi*4 ..];  K32-blocked: slab h*4 + d, [slab][row][qd*8 + hi*4 ..]
      uint16_t* op = out_k32_rows ? out + ((int64_t)(h * 4) * out_k32_rows + seq_start + qrow) * 32 + hi * 4
                                  : out + (int64_t)(seq_start + qrow) * ldo + h * DH + hi * 4;
      const int64_t dstep = out_k32_rows ? (int64_t)out_k32_rows * 32 : 32;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          u32x2_t w;
          w[0] = pack_bf16x2(o[bq][d][qd * 4 + 0] * inv, o[bq][d][qd * 4 + 1] * inv);
          w[1] = pack_bf16x2(o[bq][d][qd * 4 + 2] * inv, o[bq][d][qd * 4 + 3] * inv);
          *reinterpret_cast<u32x2_t*>(op + d * dstep + qd * 8) = w;
        }
    }
  }
}


#ifdef OMNI_DEV
#include "dev/attention_family_pipe16_mfma16x16x32.inc"
#endif

}  // namespace

namespace {
int attn_variant() {
  // dev knob: OMNI_ATTN_NQ = 1 (default; 32 queries / wave, 839-875 TF/s) or 2 (64 queries / wave).  NQ = 2 is correct
  // but measures 346 TF/s as compiled by hipcc: with ~400 live registers it shuffles ~250 values between the AGPR
  // and VGPR halves every tile and spills 19 — it needs hand-placed registers before it can pay off.
  static int v = -1;
  if (v < 0) {
    v = omni_dev_env_int("OMNI_ATTN_NQ", 1);
    if (v != 1 && v != 2) v = 1;
  }
  return v;
}
bool attn_pipelined() {
  // dev knob: OMNI_ATTN_PIPE=0 selects the unskewed register-staged kernels (then OMNI_ATTN_NQ picks 32/64 queries per wave)
  static int v = -1;
  if (v < 0) {
    v = omni_dev_env_int("OMNI_ATTN_PIPE", 1);
  }
  return v != 0;
}
int attn_block_order() {
  // dev knob: OMNI_ATTN_BLOCK_ORDER = 1 (default, XCD-aware head-major) | 0 (heads fastest)
  static const int v = omni_dev_env_int("OMNI_ATTN_BLOCK_ORDER", 1);
  return v;
}
int attn_pipe_waves(int n_heads_total, int max_seqlen) {
  // 8 waves per workgroup (256 queries, half the DMA per query: +3.7 % at B=6) once the grid is at least 6 rounds of 256
  // CUs deep; 4 waves (128 queries, finer tail) below that (+3.7 % at B=2).  dev knob: OMNI_ATTN_WAVES = 4 | 8.
  static int v = -1;
  if (v < 0) {
    v = omni_dev_env_int("OMNI_ATTN_WAVES", 0);
    if (v != 4 && v != 8) v = 0;
  }
  if (v) return v;
  const long wgs8 = (long)n_heads_total * ((max_seqlen + 255) / 256);
  return wgs8 >= 6 * 256 ? 8 : 4;
}
template <int NW, int NQ = 1, int PP = 0>
int launch_pipe(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out, int64_t ldq, int64_t ldk,
                int64_t ldv, int64_t ldo, const int32_t* cu_seqlens, int32_t B, int32_t H, int32_t max_seqlen,
                float softmax_scale, int out_k32_rows, hipStream_t s, const int32_t* item_skip = nullptr, int q_prescaled = 0) {
  static std::atomic<uint64_t> attr_done{0};     // per device (common.h omni_once_per_device)
  OMNI_TRY_STATUS(omni_once_per_device(attr_done, [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_fwd_pipe_kernel<NW, NQ, PP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
  }));
  const int qblocks = (max_seqlen + 32 * NW * NQ - 1) / (32 * NW * NQ);
  const int nh = B * H;
  hipLaunchKernelGGL((flash_attn_fwd_pipe_kernel<NW, NQ, PP>), dim3(nh * qblocks), dim3(NW * 64), LDS_BYTES, s, q, k, v, out, ldq,
                     ldk, ldv, ldo, cu_seqlens, nh, H, softmax_scale * 1.4426950408889634f, out_k32_rows, attn_block_order(), item_skip,
                     q_prescaled);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}
#ifdef OMNI_DEV
#include "dev/attention_dev_launch_pipe16.inc"
#endif
#ifdef OMNI_DEV
#include "dev/attention_dev_launch_round1.inc"
#endif
}  // namespace



int omni_internal_flash_attn(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out, int64_t ldq,
                             int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* cu_seqlens, int32_t B, int32_t H,
                             int32_t head_dim, int32_t max_seqlen, float softmax_scale, int32_t out_k32_rows,
                             const int32_t* item_skip, int32_t q_prescaled, void* part_ws, size_t part_ws_bytes, void* stream) {
  if (!q || !k || !v || !out || !cu_seqlens || B <= 0 || H <= 0 || max_seqlen <= 0 || out_k32_rows < 0)
    return OMNI_ERR_BAD_ARG;
  if (head_dim != DH) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(q) || !omni_aligned16(k) || !omni_aligned16(v) || (reinterpret_cast<uintptr_t>(out) & 7) ||
      (ldq % 8) || (ldk % 8) || (ldv % 8) || (!out_k32_rows && (ldo % 4)))
    return OMNI_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
#if OMNI_ATTN_W64
  // >= 2 rounds of 256-query workgroups over the 256 CUs: the 64-queries-per-wave kernel (attention_w64.hip); smaller grids
  // keep the finer 128 / 256-query blocks of the two-waves-per-SIMD kernel below
  if ((long)B * H * ((max_seqlen + 255) / 256) >= 512 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
      (out_k32_rows || ldo % 8 == 0))                       // its epilogue stores whole 16-byte pieces
    return omni_internal_flash_attn_w64(q, k, v, out, ldq, ldk, ldv, ldo, cu_seqlens, B, H, max_seqlen, softmax_scale,
                                        out_k32_rows, item_skip, q_prescaled, part_ws, part_ws_bytes, stream);
#endif
#ifdef OMNI_DEV
#include "dev/attention_dev_dispatch.inc"
#endif
  if (attn_pipe_waves(B * H, max_seqlen) == 8)
    return launch_pipe<8>(q, k, v, out, ldq, ldk, ldv, ldo, cu_seqlens, B, H, max_seqlen, softmax_scale, out_k32_rows, s, item_skip,
                          q_prescaled);
  return launch_pipe<4>(q, k, v, out, ldq, ldk, ldv, ldo, cu_seqlens, B, H, max_seqlen, softmax_scale, out_k32_rows, s, item_skip,
                        q_prescaled);
}

extern "C" int omni_flash_attn_fwd_ex(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out,
                                      int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* cu_seqlens,
                                      int32_t B, int32_t H, int32_t head_dim, int32_t max_seqlen, float softmax_scale,
                                      int32_t out_k32_rows, omni_stream stream) {
  return omni_internal_flash_attn(q, k, v, out, ldq, ldk, ldv, ldo, cu_seqlens, B, H, head_dim, max_seqlen, softmax_scale,
                                  out_k32_rows, nullptr, 0, nullptr, 0, stream);
}

extern "C" size_t omni_flash_attn_workspace_bytes(int32_t B, int32_t H) {
  if (B <= 0 || H <= 0) return 0;
  return omni_internal_flash_attn_w64_ws_bytes(B, H);
}

extern "C" int omni_flash_attn_fwd_ws(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out,
                                      int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* cu_seqlens,
                                      int32_t B, int32_t H, int32_t head_dim, int32_t max_seqlen, float softmax_scale,
                                      int32_t out_k32_rows, void* workspace, size_t workspace_bytes, omni_stream stream) {
  if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15)) return OMNI_ERR_ALIGN;
  return omni_internal_flash_attn(q, k, v, out, ldq, ldk, ldv, ldo, cu_seqlens, B, H, head_dim, max_seqlen, softmax_scale,
                                  out_k32_rows, nullptr, 0, workspace, workspace ? workspace_bytes : 0, stream);
}

extern "C" int omni_flash_attn_fwd(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out,
                                   int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* cu_seqlens,
                                   int32_t B, int32_t H, int32_t head_dim, int32_t max_seqlen, float softmax_scale,
                                   omni_stream stream) {
  return omni_flash_attn_fwd_ex(q, k, v, out, ldq, ldk, ldv, ldo, cu_seqlens, B, H, head_dim, max_seqlen,
                                softmax_scale, 0, stream);
}
