// Grouped bf16 GEMM with fused epilogues for gfx950:  Y_g = epi(A_g · W_gᵀ + bias_g).
//
// The kernels share one tile geometry (one workgroup = one 256x256 output tile, 8 waves = 2(M) x 4(N), 512 threads,
// 1 WG / CU) and one MFMA scheme.  The product library holds TWO of them:
//   3  gemm_bf16_pp_kernel   — the production kernel (default): BK = 64 K-tiles, the two wave groups of the workgroup run
//      half a phase apart (one in an MFMA-only cluster while its SIMD partner issues reads and LDS-DMA), see its comment;
//   1  gemm_bf16_ring_kernel — round 1's kernel, now the fallback for what the ping-pong kernel does not take (K % 64 != 0,
//      operand offsets beyond 32 bits, outputs that cannot use the row-coalesced epilogue); bit-identical results.
// Everything else that was built and measured against them — the first design (family 0), the hipcc-scheduled 4-wave kernel (2),
// the hand-placed 4-wave kernels with the accumulators in the AGPR half (4 "Q4", 7 "V4": the vendor kernel's schedule), the
// two-phase ping-pong variants (5, 6), the persistent ping-pong kernel (8), the ablation entry points and the phase probe — lives
// under dev/ as fragments that ONLY a -DOMNI_DEV build includes (tools/build_variants.sh; a family is then picked by
// omni_gemm_params.kernel_hint = 16 + family or by OMNI_GEMM_VARIANT).  The product library takes omni_gemm_params.kernel_hint =
// OMNI_GEMM_KERNEL_RING (force the fallback family) / OMNI_GEMM_KERNEL_SPLITK_TALL and nothing else: no environment variable, no
// global setter.
// Common to all:
//   Staging is direct-to-LDS (global_load_lds_dwordx4, 1 KiB per wave-instruction).  The LDS image is row-major with the
//   16-byte chunk index XOR-swizzled; because the DMA destination is lane-linear the swizzle is applied to the per-lane
//   SOURCE address and again on the ds_read_b128 address (cdna_hip_programming.md rule 21): the four 16-lane groups of a
//   ds_read_b128 hit 16 distinct 16-byte slots (0 conflicts measured).
//   MFMA: v_mfma_f32_32x32x16_bf16 with SWAPPED operands (A-operand = W fragment, B-operand = A fragment), so a wave's
//   accumulator holds Cᵀ blocks: lane (l&31) owns ONE output row and, per 4 registers, FOUR CONSECUTIVE output columns.
//   blockIdx is remapped so that each XCD (block b runs on XCD b % 8) owns a contiguous band of tiles and walks it
//   GROUP_M row-tiles at a time: neighbours share A/W panels in that XCD's private L2.
//
// Roofline: MFMA-bound.  Algorithmic work = 2*M*N*K flop per launch.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int NTHREADS = 512;
constexpr int TILE_BYTES = BM * BK * 2;      // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;   // 128 KiB
constexpr int GROUP_M_DEFAULT = 4;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#ifndef OMNI_FIN_REORDER
#define OMNI_FIN_REORDER 1        // finish kernels: 1 = a row's partial loads are issued before its map-dependent gate / residual loads
#endif                            // (with OMNI_FIN_BATCH below: same-box A/B, profiles/r06c_ab_finish_variants.log)
#ifndef OMNI_GLDS_AUX
#define OMNI_GLDS_AUX 0   // cache-policy bits of the DMA loads: 1 = sc0, 2 = nt, 16 = sc1
#endif
// The ring kernel issues its LDS-DMA from inline asm (M0 = wave-uniform LDS byte address; the hardware adds lane*16), in
// the SADDR form whenever the operand's byte offsets fit 32 bits: uniform 64-bit stage base in SGPRs + ONE 32-bit per-lane
// offset instead of a 64-bit VGPR address pair per piece.  Same-box A/B at the bench shapes: MLP-up 1175 -> 1236 TF/s,
// QKV 1235 -> 1294 (the per-piece issue cost of global_load_lds is what the k-loop is most sensitive to).  Both forms come
// from asm so that M0 has a single writer in that kernel (the builtin's M0 tracking would not see the asm's writes).
OMNI_DEVINL void glds16_saddr(const char* base, uint32_t lane_byte_off, uint32_t lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds_byte_addr), "v"(lane_byte_off), "s"(base) : "memory");
}
OMNI_DEVINL void glds16_vaddr(const void* gsrc, uint32_t lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(lds_byte_addr), "v"(gsrc) : "memory");
}
OMNI_DEVINL void glds16(const void* gsrc, uint32_t lds_byte_addr) {
  // wave-uniform LDS base (goes to M0); the hardware adds lane*16.
  __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)(uintptr_t)lds_byte_addr, 16, 0, OMNI_GLDS_AUX);
}

// ---- hand-pipelined fragment reads -------------------------------------------------------------------------
// hipcc always retires a group of ds_reads with lgkmcnt(0) and refuses to keep the NEXT k-step's reads in flight
// behind the current k-step's MFMAs (it sinks them back or waits for all of them).  With 8 waves leaving the
// barrier together that makes every k-step "burst-read 48 KiB, wait, then MFMA": the matrix pipe idles ~50 %.
// So the reads are issued from inline asm (invisible to the compiler's waitcnt pass) and retired with COUNTED
// waits: two k-steps of fragments (12 x ds_read_b128) are in flight while 8 MFMAs run.
// Rules followed (cdna_hip_programming.md §5.7 form iii, rule 18): every asm load is "=v", the wait is its own
// statement, and a sched_barrier(0) fences the consuming MFMAs below the wait.
template <int OFF, bool SKIP = false>
OMNI_DEVINL bf16x8_t lds_read16(uint32_t addr) {
  bf16x8_t v;
  if (SKIP) asm volatile("" : "=v"(v) : "v"(addr));
  else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}

// One LDS stage of NKS k-steps (16 k each) for a wave tile of 128 (4 x 32 rows of A) x 64 (2 x 32 rows of W).
// a_base[ks] / w_base[ks]: per-lane byte offsets of the fragment rows for k-step ks (swizzle folded in);
// MB_STRIDE = bytes between 32-row blocks, W_BASE = byte offset of the W image inside a stage.
// ABL (dev-only ablation, tools/bench_ablate.py): 0 normal, 2 no MFMA, 3/5 no fragment reads
template <int NKS, int MB_STRIDE, int W_BASE, int ABL, typename Hook>
OMNI_DEVINL void mma_stage(f32x16_t (&acc)[2][4], const uint32_t (&a_base)[NKS], const uint32_t (&w_base)[NKS],
                           uint32_t stage_addr, Hook&& after_kstep) {
  bf16x8_t wf[2][2], af[2][4];
  if (ABL == 3 || ABL == 5) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int i = 0; i < 2; ++i) asm volatile("" : "=v"(wf[b][i]));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(af[b][i]));
    }
  }
#define OMNI_READ_KS(buf, ks)                                        \
  do {                                                               \
    if (ABL == 3 || ABL == 5) break;                                 \
    const uint32_t aa_ = a_base[ks] + stage_addr;                    \
    const uint32_t wa_ = w_base[ks] + stage_addr;                    \
    wf[buf][0] = lds_read16<W_BASE>(wa_);                            \
    wf[buf][1] = lds_read16<W_BASE + MB_STRIDE>(wa_);                \
    af[buf][0] = lds_read16<0>(aa_);                                 \
    af[buf][1] = lds_read16<MB_STRIDE>(aa_);                         \
    af[buf][2] = lds_read16<2 * MB_STRIDE>(aa_);                     \
    af[buf][3] = lds_read16<3 * MB_STRIDE>(aa_);                     \
  } while (0)
  OMNI_READ_KS(0, 0);
  if (NKS > 1) OMNI_READ_KS(1, 1);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if (ks + 1 < NKS) {
      asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ABL == 2) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) asm volatile("" ::"v"(wf[ks & 1][nb]));
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) asm volatile("" ::"v"(af[ks & 1][mb]));
    } else {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][nb], af[ks & 1][mb], acc[nb][mb], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 2 < NKS) {
      if ((ks & 1) == 0) OMNI_READ_KS(0, (ks + 2 < NKS ? ks + 2 : 0));
      else OMNI_READ_KS(1, (ks + 2 < NKS ? ks + 2 : 0));
    }
    // DMA issue for a later stage is SPREAD over the k-steps (each global_load_lds costs the issuing wave
    // 60-185 issue cycles: a burst of 8 right after the barrier stalls every wave while the matrix pipe idles)
    after_kstep(ks);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef OMNI_READ_KS
}

// acc[nb][mb][4q+j] = C[m][n],  m = mrow0 + mb*32 + l31,  n = ncol0 + nb*32 + 8q + 4hi + j
// Group selection with STATIC kernarg indexing.  `P.g[gi]` with a runtime gi makes the compiler copy the whole
// by-value kernarg struct to scratch and re-load its fields from there (scratch_load in the epilogue loops); a
// field-wise uniform select keeps everything in SGPRs.
OMNI_DEVINL omni_gemm_group pick_group(const omni_gemm_params& P, int gi) {
  omni_gemm_group G;
#define OMNI_PICK(f) G.f = gi ? P.g[1].f : P.g[0].f
  OMNI_PICK(A); OMNI_PICK(lda); OMNI_PICK(a_row_map); OMNI_PICK(M); OMNI_PICK(W); OMNI_PICK(bias); OMNI_PICK(out);
  OMNI_PICK(out1); OMNI_PICK(out2); OMNI_PICK(ldo); OMNI_PICK(out_row_map); OMNI_PICK(res); OMNI_PICK(ldres);
  OMNI_PICK(gate); OMNI_PICK(gate_item_stride); OMNI_PICK(row_item_map); OMNI_PICK(rows_per_item);
  OMNI_PICK(a_k32_rows); OMNI_PICK(out_k32_rows);
  OMNI_PICK(qk_norm_q_w); OMNI_PICK(qk_norm_k_w); OMNI_PICK(qk_rope_cos); OMNI_PICK(qk_rope_sin); OMNI_PICK(qk_row_pos);
  OMNI_PICK(qk_eps); OMNI_PICK(qk_q_scale); OMNI_PICK(tile_skip); OMNI_PICK(a_scale); OMNI_PICK(w_scale);
#undef OMNI_PICK
  return G;
}

template <int EPI, int NB, int MB>
OMNI_DEVINL void gemm_epilogue_t(const omni_gemm_params& P, const omni_gemm_group& G, f32x16_t (&acc)[NB][MB],
                                 int mrow0, int ncol0, int l31, int hi) {
  const int M = G.M, N = P.N;
  // ---- epilogue: acc[nb][mb][4q+j] = C[m][n],  m = m0+wm*128+mb*32+l31,  n = n0+wn*64+nb*32+8q+4hi+j
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mrow0 + mb * 32 + l31;
    if (m >= M) continue;
    const int64_t orow = G.out_row_map ? G.out_row_map[m] : m;
    int item = 0;
    if (EPI == OMNI_EPI_BIAS_GATE_RES) item = G.row_item_map ? G.row_item_map[m] : m / G.rows_per_item;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = ncol0 + nb * 32 + q * 8 + hi * 4;
        if (n >= N) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[nb][mb][q * 4 + j];
        if (G.bias) {
          const u32x2_t b = *reinterpret_cast<const u32x2_t*>(G.bias + n);
          v[0] += bf16_lo(b[0]); v[1] += bf16_hi(b[0]); v[2] += bf16_lo(b[1]); v[3] += bf16_hi(b[1]);
        }
        uint16_t* dst;
        if (EPI == OMNI_EPI_BIAS_SPLIT3 || EPI == OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE) {
          const int which = n / P.split_n;
          uint16_t* base = which == 0 ? G.out : (which == 1 ? G.out1 : G.out2);
          dst = base + orow * G.ldo + (n - which * P.split_n);
        } else {
          dst = G.out + orow * G.ldo + n;
        }
        if (EPI == OMNI_EPI_BIAS_GELU_TANH) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = gelu_tanh_f(v[j]);
        }
        if (EPI == OMNI_EPI_BIAS_GATE_RES) {
          const u32x2_t g = *reinterpret_cast<const u32x2_t*>(G.gate + (int64_t)item * G.gate_item_stride + n);
          const u32x2_t r = *reinterpret_cast<const u32x2_t*>(G.res + orow * G.ldres + n);
          v[0] = bf16_lo(r[0]) + bf16_lo(g[0]) * v[0];
          v[1] = bf16_hi(r[0]) + bf16_hi(g[0]) * v[1];
          v[2] = bf16_lo(r[1]) + bf16_lo(g[1]) * v[2];
          v[3] = bf16_hi(r[1]) + bf16_hi(g[1]) * v[3];
        }
        u32x2_t o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2_t*>(dst) = o;
      }
    }
  }
}

template <int EPI>
OMNI_DEVINL void gemm_epilogue(const omni_gemm_params& P, const omni_gemm_group& G, f32x16_t (&acc)[2][4], int m0,
                               int n0, int wm, int wn, int l31, int hi) {
  gemm_epilogue_t<EPI, 2, 4>(P, G, acc, m0 + wm * 128, n0 + wn * 64, l31, hi);
}

// ------------------------------------------------------------------------------------------------
// Row-coalesced epilogue for the 8-wave (2M x 4N) kernels.  The MFMA C layout gives a lane 4 consecutive columns of
// 32 DIFFERENT rows, so a direct store instruction touches 32 cache lines with 16 B each; tools/bench_ksweep.py puts
// that at ~10 us per 256x256 tile (8192 partial-line writes; another ~10 us for the residual reads of GATE_RES) —
// 10-20 % of a K=3072 tile.  Here the tile is first written to LDS (free once the operand ring is drained) as bf16
// [256][256] with a 16-B row pad (ds_write_b64, 2-way = optimal bank use), then every thread moves 16 B of one row:
// a wave instruction covers 2 rows x 512 contiguous bytes (8 full lines) for the stores AND the residual/gate loads.
// GATE_RES note: acc+bias is rounded to bf16 in LDS before res + gate*x — exactly the rounding point of the
// reference's bf16 nn.Linear output (qwen_image_transformer.py: `hidden_states + gate * attn_output`).
// ------------------------------------------------------------------------------------------------
#ifndef OMNI_EPI_STORE_POLICY_ID
#define OMNI_EPI_STORE_POLICY_ID 1   // cache-policy bits of the output stores: 0 none, 1 nt, 2 sc0 sc1, 3 sc1.  Same-box A/B
                                     // at the bench QKV shape: 1176 / 1160 / 1180 / 1184 us; bench 0.3709 -> 0.3725 images/s with nt
#endif
#if OMNI_EPI_STORE_POLICY_ID == 1
#define OMNI_EPI_STORE_POLICY " nt"
#elif OMNI_EPI_STORE_POLICY_ID == 2
#define OMNI_EPI_STORE_POLICY " sc0 sc1"
#elif OMNI_EPI_STORE_POLICY_ID == 3
#define OMNI_EPI_STORE_POLICY " sc1"
#else
#define OMNI_EPI_STORE_POLICY ""
#endif
constexpr int EPI_LDS_STRIDE = BN * 2 + 16;              // 528 B per C row in LDS
constexpr int EPI_LDS_BYTES = BM * EPI_LDS_STRIDE;       // 132 KiB

// tag: phase 2 reads the bf16 C tile that phase 1 staged in LDS
struct EpiFromLds {};
// split-K finish (gemm_splitk_finish_kernel): phase 2 forms C = sum over the K-splits of the fp32 partial tiles (in split
// order: deterministic) + bias (+ GELU), rounded to bf16 exactly where the fused kernel rounds (acc + bias -> bf16 in LDS)
struct EpiFromPartials {
  const float* ws;          // whole-launch split-K: [nsplit][Mtot][N] fp32;  tail split: this tile's [nsplit][256][256]
  int64_t split_stride;     // Mtot * N                                       |  256 * 256
  int64_t row_base;         // first row of this group inside the Mtot rows   |  0
  int nsplit;
  int b_begin, b_end;       // row batches (4 rows per thread x NT / 32 rows in flight: 64 rows at 512 threads, 32 at 256) this workgroup finishes
  int64_t row_stride;       // N                                              |  256
  int row0, col0;           // 0, 0                                           |  m0, n0 of the tile (partials are tile-local)
};

template <int EPI, typename WriteTile, typename CSrc = EpiFromLds, int NT = NTHREADS, int BATCH_ = 4>
OMNI_DEVINL void gemm_epilogue_lds_impl(const omni_gemm_params& P, const omni_gemm_group& G, int m0, int n0, char* smem,
                                        int tid, WriteTile write_tile, CSrc csrc = CSrc{}) {
  constexpr bool FROM_PARTIALS = std::is_same<CSrc, EpiFromPartials>::value;
  const int M = G.M, N = P.N;
  // Phase 2 moves rows in batches of 4 per thread: batch b = tile rows (4b+j)*16 + rsub.  Its pipeline is
  //   row-map loads (b+2)  |  residual/gate/LDS loads (b+1)  |  math + stores (b)
  // in a ROLLED loop: fully unrolled, hipcc hoists all 16 rows' 64-bit addresses and predicates above phase 1, where
  // they are live together with the 128 accumulator registers and spill.
  constexpr int RS = NT / 32;              // rows in flight per pass: a row is written by 32 threads (16 B each)
  constexpr int BATCH = BATCH_, NBATCH = BM / RS / BATCH;
  constexpr bool SPLIT = EPI == OMNI_EPI_BIAS_SPLIT3 || EPI == OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE;
  constexpr bool QKROPE = EPI == OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE;
  struct RowIdx { int ro[BATCH], im[BATCH], ps[BATCH]; };
  struct RowData { u32x4_t c[BATCH], g[BATCH], r[BATCH]; u32x2_t cw[BATCH], sw[BATCH]; int ro[BATCH]; };
  const int rsub = tid >> 5;
  const bool v_tile = QKROPE && n0 >= 2 * P.split_n;      // the V third of the fused QKV output: no norm, no RoPE
  const float inv_rpi = 1.0f / (float)max(G.rows_per_item, 1);
  auto load_maps = [&](int b, RowIdx& x) {
    int mc[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      mc[j] = min(m0 + (b * BATCH + j) * RS + rsub, M - 1);     // rows past M: clamped here, masked at the store
      x.ro[j] = mc[j];
    }
    if (G.out_row_map) {                 // uniform branches around whole groups of loads, none between the loads
#pragma unroll
      for (int j = 0; j < BATCH; ++j) x.ro[j] = G.out_row_map[mc[j]];
    }
    if (QKROPE && !v_tile) {
#pragma unroll
      for (int j = 0; j < BATCH; ++j) x.ps[j] = G.qk_row_pos[mc[j]];     // RoPE table row of the LOGICAL row
    }
    if (EPI == OMNI_EPI_BIAS_GATE_RES) {
      if (G.row_item_map) {
#pragma unroll
        for (int j = 0; j < BATCH; ++j) x.im[j] = G.row_item_map[mc[j]];
      } else {
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {        // exact m / rows_per_item from a float estimate + one fix-up (the
          int q = (int)((float)mc[j] * inv_rpi);   // generic integer division needs ~10 temporaries per row)
          const int rem = mc[j] - q * G.rows_per_item;
          q += (rem >= G.rows_per_item) - (rem < 0);
          x.im[j] = q;
        }
      }
    }
  };
  RowIdx x0, x1;
  int bb = 0, be = NBATCH;               // row batches of this workgroup (all four, except in the split-K finish)
  if constexpr (FROM_PARTIALS) { bb = csrc.b_begin; be = csrc.b_end; }
  load_maps(bb, x0);                     // in flight under phase 1
  if (bb + 1 < be) load_maps(bb + 1, x1);

  if (!FROM_PARTIALS) {
    __builtin_amdgcn_s_barrier();        // every wave has finished reading the operand ring
    write_tile();                        // accumulators (+ GELU) -> bf16 C tile in LDS, layout-specific
    __syncthreads();
  }
  const int chunk = tid & 31;
  const int n = n0 + chunk * 8;
  if (n >= N) return;
  uint16_t* obase = G.out;
  int ncol_out = n;
  int which = 0;
  if (SPLIT) {
    which = n / P.split_n;
    const uintptr_t o0 = (uintptr_t)G.out, o1 = (uintptr_t)G.out1, o2 = (uintptr_t)G.out2;   // integer selects: a
    uintptr_t ob = which == 1 ? o1 : o0;                    // pointer select chain is turned into an indexed
    ob = which == 2 ? o2 : ob;                              // load from a scratch copy of G
    obase = reinterpret_cast<uint16_t*>(ob);
    ncol_out = n - which * P.split_n;
  }
  const char* lds_row = FROM_PARTIALS ? nullptr : smem + rsub * EPI_LDS_STRIDE + chunk * 16;
  u32x4_t bias8 = {0u, 0u, 0u, 0u};                  // finish kernels: this thread's 8 bias values, loaded ONCE (not per row)
  if (FROM_PARTIALS && G.bias) bias8 = *reinterpret_cast<const u32x4_t*>(G.bias + n);
  auto load_data = [&](int b, const RowIdx& x, RowData& d) {
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      d.ro[j] = x.ro[j];
      if (EPI == OMNI_EPI_BIAS_GATE_RES && !(FROM_PARTIALS && OMNI_FIN_REORDER)) {
        d.g[j] = *reinterpret_cast<const u32x4_t*>(G.gate + (int64_t)x.im[j] * G.gate_item_stride + n);
        d.r[j] = *reinterpret_cast<const u32x4_t*>(G.res + (int64_t)x.ro[j] * G.ldres + n);
      }
      if constexpr (FROM_PARTIALS) {
        const int64_t row = csrc.row_base + min(m0 + (b * BATCH + j) * RS + rsub, M - 1) - csrc.row0;
        const float* src = csrc.ws + row * csrc.row_stride + (n - csrc.col0);
        // All of a row's partials (up to 8 splits x 32 B) are requested before the first add: one load per loop trip (round 2's
        // form) left a thread with 32 B in flight, and the finish of a 6-way split at 640 rows x 3072 columns ran at 3.2 TB/s
        // (14.6 us, profiles/r06_first_profiles_step_shapes.txt).  Splits past nsplit re-read the last one (unconditional loads
        // can be hoisted) and are skipped in the sum; the order of the adds stays the split order (deterministic, as before).
        // Round 6, third session: the request width follows the split factor (2, 3, 4, 6 exactly; else 8 at a time) — with the fixed
        // 8 the finish of a 2-way split (QKV of one 256^2 CFG pair) issued four loads for every one it needed and ran 18.5 us
        // against 14.3 us for the same bytes 6-way.  Same adds, same order.
        f32x4_t lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
        auto sum_splits = [&](auto wc) {
          constexpr int W = decltype(wc)::value;
          for (int sp0 = 0; sp0 < csrc.nsplit; sp0 += W) {
            f32x4_t pl[W], ph[W];
#pragma unroll
            for (int u = 0; u < W; ++u) {
              const float* q = src + (int64_t)min(sp0 + u, csrc.nsplit - 1) * csrc.split_stride;
              pl[u] = *reinterpret_cast<const f32x4_t*>(q);
              ph[u] = *reinterpret_cast<const f32x4_t*>(q + 4);
            }
#pragma unroll
            for (int u = 0; u < W; ++u)
              if (sp0 + u < csrc.nsplit) { lo += pl[u]; hi += ph[u]; }
          }
        };
        switch (csrc.nsplit) {                          // uniform over the launch
          case 2: sum_splits(std::integral_constant<int, 2>{}); break;
          case 3: sum_splits(std::integral_constant<int, 3>{}); break;
          case 4: sum_splits(std::integral_constant<int, 4>{}); break;
          case 6: sum_splits(std::integral_constant<int, 6>{}); break;
          default: sum_splits(std::integral_constant<int, 8>{}); break;
        }
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (G.bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] += bf16_lo(bias8[e]); v[2 * e + 1] += bf16_hi(bias8[e]); }
        }
        if (EPI == OMNI_EPI_BIAS_GELU_TANH) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) d.c[j][e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        if (EPI == OMNI_EPI_BIAS_GATE_RES && OMNI_FIN_REORDER) {
          d.g[j] = *reinterpret_cast<const u32x4_t*>(G.gate + (int64_t)x.im[j] * G.gate_item_stride + n);
          d.r[j] = *reinterpret_cast<const u32x4_t*>(G.res + (int64_t)x.ro[j] * G.ldres + n);
        }
      } else {
        d.c[j] = *reinterpret_cast<const u32x4_t*>(lds_row + (b * BATCH + j) * RS * EPI_LDS_STRIDE);
      }
      if (QKROPE && which < 2) {   // cos / sin of the 4 rotation pairs this lane holds (sub = lane's 16-B chunk within its head);
                                   // `which` is uniform over the workgroup (split_n is a multiple of the tile width): V tiles skip
        d.cw[j] = *reinterpret_cast<const u32x2_t*>(G.qk_rope_cos + (int64_t)x.ps[j] * 64 + (chunk & 15) * 4);
        d.sw[j] = *reinterpret_cast<const u32x2_t*>(G.qk_rope_sin + (int64_t)x.ps[j] * 64 + (chunk & 15) * 4);
      }
    }
  };
  // q / k heads: RMSNorm(128) weight of this lane's 8 columns (a head = 16 consecutive lanes; which is uniform per head)
  float qkw[8];
  if (QKROPE) {
    const u32x4_t wv = *reinterpret_cast<const u32x4_t*>((which == 1 ? G.qk_norm_k_w : G.qk_norm_q_w) + (chunk & 15) * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { qkw[2 * e] = bf16_lo(wv[e]); qkw[2 * e + 1] = bf16_hi(wv[e]); }
    if (which == 0 && G.qk_q_scale != 0.0f) {      // q pre-multiplied for the attention kernel (omni_gemm_group.qk_q_scale)
#pragma unroll
      for (int e = 0; e < 8; ++e) qkw[e] *= G.qk_q_scale;
    }
  }
  RowData d0, d1;
  load_data(bb, x0, d0);
#pragma unroll 1
  for (int b = bb; b < be; ++b) {
    if (b + 1 < be) load_data(b + 1, x1, d1);
    if (b + 2 < be) load_maps(b + 2, x1);
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      u32x4_t o = d0.c[j];
      if (QKROPE && which < 2) {
        // identical arithmetic (and order) to qk_norm_rope_kernel on the bf16-rounded linear output
        float f[8], r8[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[2 * e] = bf16_lo(o[e]); f[2 * e + 1] = bf16_hi(o[e]); }
        const float cc[4] = {bf16_lo(d0.cw[j][0]), bf16_hi(d0.cw[j][0]), bf16_lo(d0.cw[j][1]), bf16_hi(d0.cw[j][1])};
        const float sn[4] = {bf16_lo(d0.sw[j][0]), bf16_hi(d0.sw[j][0]), bf16_lo(d0.sw[j][1]), bf16_hi(d0.sw[j][1])};
        qk_norm_rope_lane(f, qkw, cc, sn, G.qk_eps, r8);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(r8[2 * e], r8[2 * e + 1]);
      }
      if (EPI == OMNI_EPI_BIAS_GATE_RES) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = pack_bf16x2(bf16_lo(d0.r[j][e]) + bf16_lo(d0.g[j][e]) * bf16_lo(d0.c[j][e]),
                             bf16_hi(d0.r[j][e]) + bf16_hi(d0.g[j][e]) * bf16_hi(d0.c[j][e]));
      }
      uint16_t* dst = G.out_k32_rows
                          ? obase + ((int64_t)(ncol_out >> 5) * G.out_k32_rows + d0.ro[j]) * 32 + (ncol_out & 31)
                          : obase + (int64_t)d0.ro[j] * G.ldo + ncol_out;
      // The store is issued from inline asm: `res` may alias `out` (in-place residual), and for a compiler-visible
      // store hipcc drains vmcnt(0) before the next loads although a thread never re-reads a row it has written.
      if (m0 + (b * BATCH + j) * RS + rsub < M)
        // (+ wait states: a 128-bit store reads its data VGPRs after issue, and hipcc does not know this statement is a store —
        // the hazard was observed in gemm_epilogue_direct_k32, whose next instruction re-used the data registers)
        asm volatile("global_store_dwordx4 %0, %1, off" OMNI_EPI_STORE_POLICY "\n\ts_nop 1" ::"v"(dst), "v"(o));
    }
    d0 = d1;
  }
}

// 32x32x16 accumulator layout: acc[nb][mb][4q+j] = C[wm*128 + mb*32 + l31][wn*64 + nb*32 + 8q + 4hi + j]
template <int EPI>
OMNI_DEVINL void gemm_epilogue_lds(const omni_gemm_params& P, const omni_gemm_group& G, f32x16_t (&acc)[2][4], int m0,
                                   int n0, int wm, int wn, int l31, int hi, char* smem, int tid) {
  gemm_epilogue_lds_impl<EPI>(P, G, m0, n0, smem, tid, [&]() {
    const int ncol = wn * 64 + hi * 4;   // tile-local column of this lane's first value
    // (the bias is already in the accumulators: see the ring kernel's accumulator initialisation)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      char* rowp = smem + (wm * 128 + mb * 32 + l31) * EPI_LDS_STRIDE + ncol * 2;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = acc[nb][mb][q * 4 + j];
            if (EPI == OMNI_EPI_BIAS_GELU_TANH) v[j] = gelu_tanh_f(v[j]);
          }
          u32x2_t o;
          o[0] = pack_bf16x2(v[0], v[1]);
          o[1] = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<u32x2_t*>(rowp + (nb * 32 + q * 8) * 2) = o;
        }
    }
  });
}
// 16x16x32 accumulator layout: acc[nb][mb][j] = C[wm*128 + mb*16 + l15][wn*64 + nb*16 + 4g + j], l15 = lane & 15, g = lane >> 4.
// A ds_write_b64 of the wave covers 16 rows x 4 column groups: bank = (4*l15 + 2*g + {0,1}) mod 64 -> 2-way = the minimum
// for 512 B per instruction.
template <int EPI>
OMNI_DEVINL void gemm_epilogue_lds(const omni_gemm_params& P, const omni_gemm_group& G, f32x4_t (&acc)[4][8], int m0,
                                   int n0, int wm, int wn, int l15, int g, char* smem, int tid) {
  gemm_epilogue_lds_impl<EPI>(P, G, m0, n0, smem, tid, [&]() {
    const int ncol = wn * 64 + g * 4;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      char* rowp = smem + (wm * 128 + mb * 16 + l15) * EPI_LDS_STRIDE + ncol * 2;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[nb][mb][j];
          if (EPI == OMNI_EPI_BIAS_GELU_TANH) v[j] = gelu_tanh_f(v[j]);
        }
        u32x2_t o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2_t*>(rowp + nb * 32) = o;
      }
    }
  });
}

// Direct epilogue for K32-BLOCKED outputs without a row map (the MLP-up / GELU launch: 27 % of the DiT flops, the roofline
// kernel).  In the blocked layout [N/32][R][32] the 16 rows of an accumulator block are 16 x 64 B = 1 KiB CONTIGUOUS, and a
// lane's four values are 8 contiguous bytes of its row: the wave's store instruction for (mb, nb) covers half of that KiB
// (32 B of each of 16 consecutive rows) and the one for nb ^ 1, issued right behind it, the other half — whole lines without
// the bf16 C tile in LDS, its two barriers and the second pass (13.3 us -> a few us of the ~90 us a tile takes).
template <int EPI, int FP8>
OMNI_DEVINL void gemm_epilogue_direct_k32(const omni_gemm_params& P, const omni_gemm_group& G, f32x4_t (&acc)[4][8], int m0,
                                          int n0, int wm, int wn, int l15, int g) {
  const int M = G.M, N = P.N;
  float sw[4][4], bi[4][4];
  if (FP8) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = n0 + wn * 64 + nb * 16 + g * 4;
      f32x4_t w4 = {0.f, 0.f, 0.f, 0.f};
      u32x2_t b = {0u, 0u};
      if (n < N) {
        w4 = *reinterpret_cast<const f32x4_t*>(G.w_scale + n);
        if (G.bias) b = *reinterpret_cast<const u32x2_t*>(G.bias + n);
      }
      sw[nb][0] = w4[0]; sw[nb][1] = w4[1]; sw[nb][2] = w4[2]; sw[nb][3] = w4[3];
      bi[nb][0] = bf16_lo(b[0]); bi[nb][1] = bf16_hi(b[0]); bi[nb][2] = bf16_lo(b[1]); bi[nb][3] = bf16_hi(b[1]);
    }
  }
  uint16_t* const obase = G.out;
  const int64_t R = G.out_k32_rows;
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    const int row = m0 + wm * 128 + mb * 16 + l15;
    float sa = 1.0f;
    if (FP8) {
      int ar = min(row, M - 1);
      if (G.a_row_map) ar = G.a_row_map[ar];
      sa = G.a_scale[ar];
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {                 // one 32-column slab = the accumulator blocks nb = 2 sl, 2 sl + 1
      uint32_t a[2], b[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int nb = 2 * sl + h;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = FP8 ? acc[nb][mb][j] * sa * sw[nb][j] + bi[nb][j] : acc[nb][mb][j];   // bf16: the bias is in the accumulator
          if (EPI == OMNI_EPI_BIAS_GELU_TANH) v[j] = gelu_tanh_f(v[j]);
        }
        (h ? b : a)[0] = pack_bf16x2(v[0], v[1]);
        (h ? b : a)[1] = pack_bf16x2(v[2], v[3]);
      }
      // lane (l15, g) holds columns 4g..4g+3 of block 2sl (a) and of block 2sl+1 (b).  v_permlane16_swap exchanges a's odd
      // 16-lane rows with b's even ones: afterwards an even-g lane owns 8 consecutive columns of block 2sl (its own four and
      // its right neighbour's), an odd-g lane 8 consecutive columns of block 2sl+1 — 16 contiguous bytes each, and the wave's
      // single store covers the 16 rows' 64-B slab pieces completely: one contiguous KiB.
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
      // even g: (a = own block-2sl cols 4g.., b = neighbour g+1's block-2sl cols);  odd g: (a = neighbour g-1's block-2sl+1 cols, b = own)
      const u32x4_t o = {a[0], a[1], b[0], b[1]};
      const int n = n0 + wn * 64 + sl * 32 + (g & 1) * 16 + (g >> 1) * 8;
      uint16_t* dst = obase + ((int64_t)(n >> 5) * R + row) * 32 + (n & 31);
      // the wait states behind the store are part of the statement: a 128-bit store reads its data registers a few cycles after
      // issue, hipcc cannot see that this asm is a store and re-uses them at once as GELU temporaries (observed: dword 0 of
      // the piece = the next block's exponent argument)
      if (row < M && n < N)
        asm volatile("global_store_dwordx4 %0, %1, off" OMNI_EPI_STORE_POLICY "\n\ts_nop 2" ::"v"(dst), "v"(o) : "memory");
    }
  }
}

// fp8: C = acc * a_scale[stored row] * w_scale[col] + bias[col]  (fp32), then as the bf16 kernel (GELU, bf16 rounding, LDS tile)
template <int EPI>
OMNI_DEVINL void gemm_epilogue_lds_fp8(const omni_gemm_params& P, const omni_gemm_group& G, f32x4_t (&acc)[4][8], int m0,
                                       int n0, int wm, int wn, int l15, int g, char* smem, int tid) {
  gemm_epilogue_lds_impl<EPI>(P, G, m0, n0, smem, tid, [&]() {
    const int ncol = wn * 64 + g * 4;
    float sw[4][4], bi[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = n0 + ncol + nb * 16;
      f32x4_t w4 = {0.f, 0.f, 0.f, 0.f};
      u32x2_t b = {0u, 0u};
      if (n < P.N) {
        w4 = *reinterpret_cast<const f32x4_t*>(G.w_scale + n);
        if (G.bias) b = *reinterpret_cast<const u32x2_t*>(G.bias + n);
      }
      sw[nb][0] = w4[0]; sw[nb][1] = w4[1]; sw[nb][2] = w4[2]; sw[nb][3] = w4[3];
      bi[nb][0] = bf16_lo(b[0]); bi[nb][1] = bf16_hi(b[0]); bi[nb][2] = bf16_lo(b[1]); bi[nb][3] = bf16_hi(b[1]);
    }
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      const int rl = wm * 128 + mb * 16 + l15;
      int ar = min(m0 + rl, G.M - 1);
      if (G.a_row_map) ar = G.a_row_map[ar];
      const float sa = G.a_scale[ar];
      char* rowp = smem + rl * EPI_LDS_STRIDE + ncol * 2;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[nb][mb][j] * sa * sw[nb][j] + bi[nb][j];
          if (EPI == OMNI_EPI_BIAS_GELU_TANH) v[j] = gelu_tanh_f(v[j]);
        }
        u32x2_t o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2_t*>(rowp + nb * 32) = o;
      }
    }
  });
}

#ifdef OMNI_DEV
#include "dev/gemm_family0_first_design.inc"
#endif
// ------------------------------------------------------------------------------------------------
// Ring variant: BK = 32 per stage, 5-deep LDS ring (5 x 32 KiB = the CU's whole 160 KiB), DMA issued FOUR
// stages (= 2 BK64 tiles, ~2 us) ahead and retired with a COUNTED vmcnt(12); raw s_barrier (a __syncthreads()
// would drain the in-flight LDS-DMA with vmcnt(0)).  Motivation (profiles/r01_rocprof_summary_v1.txt): with the
// 2-stage BK=64 pipeline the MFMA pipe is busy only ~50 % of resident wave time because one tile of compute
// (2048 cycles/SIMD) is shorter than the time a 64 KiB tile needs to arrive through the 64 B/clk load path.
// LDS image per operand per stage: row-major [256][32] bf16 (64-B rows), 16-B chunk index XOR (row>>2)&3.
// One DMA wave-instruction = 16 rows x 64 B.
// ------------------------------------------------------------------------------------------------
#ifndef OMNI_RING_STAGES
#define OMNI_RING_STAGES 5
#endif
#ifndef OMNI_RING_SETPRIO
#define OMNI_RING_SETPRIO 1
#endif
#ifndef OMNI_RING_STAGGER
#define OMNI_RING_STAGGER 0   // 1: the two waves of a SIMD (wm = 0 / 1) issue their DMA pieces in DIFFERENT MFMA slots
                              // (measured -1..2 %: 1167 -> 1155 TF/s MLP-up, 1220 -> 1197 QKV, same box)
#endif
#if OMNI_RING_STAGES == 5
#define OMNI_RING_VMCNT "s_waitcnt vmcnt(8)"   // (LEAD - 2) stages x 4 DMA pieces per wave stay in flight across a barrier
#elif OMNI_RING_STAGES == 4
#define OMNI_RING_VMCNT "s_waitcnt vmcnt(4)"
#else
#error "OMNI_RING_STAGES must be 4 or 5"
#endif
constexpr int RBK = 32, RSTAGES = OMNI_RING_STAGES;
constexpr int ROP_BYTES = BM * RBK * 2;        // 16 KiB per operand per stage
constexpr int RSTAGE_BYTES = 2 * ROP_BYTES;    // 32 KiB
constexpr int RLDS_BYTES = RSTAGES * RSTAGE_BYTES;  // 160 KiB

// ABL: dev-only ablation (1 no DMA in loop, 3 no fragment reads, 4 no vmcnt/barrier).  COALESCED selects the epilogue at
// compile time: with both in one kernel their hoisted set-up code overlaps the live accumulators and spills.
template <int EPI, int ABL = 0, bool COALESCED = false, bool SADDR = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_ring_kernel(const omni_gemm_params P, int mtiles0,
                                                                       int tiles_m, int tiles_n, int GROUP_M) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Tile loop: with gridDim.x == tile count every workgroup runs it once; the PERSISTENT launch uses one workgroup per
  // CU that walks tiles bid, bid + gridDim.x, ... (gridDim.x is a multiple of 8, so a workgroup stays on its XCD's band):
  // no workgroup teardown / launch between the tiles of a CU.
  const int nwg = tiles_m * tiles_n;
#pragma unroll 1
  for (int bid = blockIdx.x; bid < nwg; bid += gridDim.x) {
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int band_sz = GROUP_M * tiles_n;
  const int band = lid / band_sz, in_band = lid - band * band_sz;
  const int first_m = band * GROUP_M;
  const int gm = min(GROUP_M, tiles_m - first_m);
  const int mt = first_m + in_band % gm;
  const int nt = in_band / gm;
  const int gi = (mt >= mtiles0) ? 1 : 0;
  const omni_gemm_group G = pick_group(P, gi);
  const int m0 = (gi ? mt - mtiles0 : mt) * BM;
  const int n0 = nt * BN;
  const int M = G.M, N = P.N, K = P.K;
  if (G.tile_skip && G.tile_skip[gi ? mt - mtiles0 : mt]) continue;   // device-side predicate (omni_teacache)

  const uint16_t* a_src[2];
  const uint16_t* w_src[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    int ar = min(m0 + r, M - 1);
    if (G.a_row_map) ar = G.a_row_map[ar];
    a_src[j] = G.A + (G.a_k32_rows ? (int64_t)ar * RBK : (int64_t)ar * G.lda) + c * 8;
    const int wr = min(n0 + r, N - 1);
    w_src[j] = G.W + (P.w_k32_blocked ? (int64_t)wr * RBK : (int64_t)wr * K) + c * 8;
  }
  // SADDR (template): the host has checked that every operand byte offset fits 32 bits (ring_saddr_ok)
  uint32_t a_off[2], w_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    a_off[j] = (uint32_t)(reinterpret_cast<const char*>(a_src[j]) - reinterpret_cast<const char*>(G.A));
    w_off[j] = (uint32_t)(reinterpret_cast<const char*>(w_src[j]) - reinterpret_cast<const char*>(G.W));
  }
  // elements between two k-stages of one W row: 32 in row-major, a whole [N][32] slab in the K32-blocked layout (where
  // the 16 rows of a DMA piece are 1 KiB contiguous -> 8 full-line requests per piece instead of 16 half-line ones)
  const int64_t wstep = P.w_k32_blocked ? (int64_t)N * RBK : RBK;
  const int64_t astep = G.a_k32_rows ? (int64_t)G.a_k32_rows * RBK : RBK;   // same for a K32-blocked A operand
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  auto issue_piece = [&](int slot, int st, int piece) {   // piece 0..3 = A0, W0, A1, W1 (1 KiB each)
    const uint32_t base = lds0 + slot * RSTAGE_BYTES + (wave * 2) * 1024;
    const int part = piece >> 1;
    if (piece & 1) {
      if (SADDR) glds16_saddr(reinterpret_cast<const char*>(G.W) + st * wstep * 2, w_off[part], base + ROP_BYTES + part * 1024);
      else glds16_vaddr(w_src[part] + st * wstep, base + ROP_BYTES + part * 1024);
    } else {
      if (SADDR) glds16_saddr(reinterpret_cast<const char*>(G.A) + st * astep * 2, a_off[part], base + part * 1024);
      else glds16_vaddr(a_src[part] + st * astep, base + part * 1024);
    }
  };
  auto issue_part = [&](int slot, int st, int part) {   // part 0..1
    issue_piece(slot, st, 2 * part);
    issue_piece(slot, st, 2 * part + 1);
  };
  auto issue_stage = [&](int slot, int st) {
    issue_part(slot, st, 0);
    issue_part(slot, st, 1);
  };

  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5;
  // per-lane fragment byte offsets per k-step: row*64 + ((ks*2+hi) ^ swz)*16, swz = (row>>2)&3 = (l31>>2)&3
  uint32_t a_base[2], w_base[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const uint32_t chunk = ((uint32_t)(ks * 2 + hi) ^ ((l31 >> 2) & 3)) << 4;
    a_base[ks] = (wm * 128 + l31) * 64 + chunk;
    w_base[ks] = (wn * 64 + l31) * 64 + chunk;
  }

  // The accumulators start at the BIAS (coalesced epilogue) instead of 0: same cost as the zero fill, and the epilogue's 128
  // adds + 128 bf16 unpacks per lane (with all 8 waves in the epilogue at once) disappear.
  f32x16_t acc[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    float bini[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wn * 64 + hi * 4 + nb * 32 + q * 8;
      u32x2_t b = {0u, 0u};
      if (COALESCED && G.bias && n < N) b = *reinterpret_cast<const u32x2_t*>(G.bias + n);
      bini[q * 4 + 0] = bf16_lo(b[0]); bini[q * 4 + 1] = bf16_hi(b[0]);
      bini[q * 4 + 2] = bf16_lo(b[1]); bini[q * 4 + 3] = bf16_hi(b[1]);
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][mb][i] = bini[i];
  }

  // ---- continuous pipeline ---------------------------------------------------------------------------------
  // global k-step g = 2*stage + ks.  Fragment reads run TWO k-steps ahead of the MFMAs and cross stage boundaries:
  // at barrier B_t every wave has already waited for its DMA pieces of stages <= t+1, so stage t+1 is visible
  // one barrier early and its first fragments are fetched while stage t's last MFMAs run.  B_t also frees the
  // slot of stage t-1 (all its reads were consumed by MFMAs issued before B_t), which is refilled with stage
  // t+4.  DMA lead = 3 stage-times (~1.5 BK64 tiles); only stages t+2, t+3 (8 DMAs per wave) stay in flight
  // across B_t (counted vmcnt(8)); the matrix pipe never waits for an LDS read burst after a barrier.
  const int nst = K / RBK;
  constexpr int LEAD = RSTAGES - 1;
#pragma unroll
  for (int st = 0; st < LEAD; ++st)
    if (st < nst) issue_stage(st, st);
  bf16x8_t wf[2][2], af[2][4];
#define OMNI_RING_ADDR(slot_, ks)                                            \
  const uint32_t sa_ = lds0 + (slot_) * RSTAGE_BYTES;                       \
  const uint32_t aa_ = a_base[ks] + sa_, wa_ = w_base[ks] + sa_;
#define OMNI_RING_READ(buf, slot_, ks)                                      \
  do {                                                                      \
    OMNI_RING_ADDR(slot_, ks)                                               \
    wf[buf][0] = lds_read16<ROP_BYTES, ABL == 3>(wa_);                                \
    wf[buf][1] = lds_read16<ROP_BYTES + 32 * 64, ABL == 3>(wa_);                      \
    af[buf][0] = lds_read16<0, ABL == 3>(aa_);                                        \
    af[buf][1] = lds_read16<32 * 64, ABL == 3>(aa_);                                  \
    af[buf][2] = lds_read16<2 * 32 * 64, ABL == 3>(aa_);                              \
    af[buf][3] = lds_read16<3 * 32 * 64, ABL == 3>(aa_);                              \
  } while (0)
// 8 MFMAs of one k-step (A-fragment-major order) with, interleaved in their issue shadows:
//   * the reads of k-step g+2 into the SAME buffer: af[mb] is dead after its two MFMAs, wf[] after the last pair;
//   * two single-piece DMA issues (DMA_A after MFMA 1, DMA_B after MFMA 5).
#define OMNI_RING_PAIR(buf, mb)                                                                            \
  acc[0][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[buf][0], af[buf][mb], acc[0][mb], 0, 0, 0);        \
  acc[1][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[buf][1], af[buf][mb], acc[1][mb], 0, 0, 0);        \
  __builtin_amdgcn_sched_barrier(0);
#define OMNI_RING_MMA(buf, PREFETCH, nslot_, ks, DMA_A, DMA_B)                                               \
  do {                                                                                                       \
    OMNI_RING_ADDR(nslot_, ks)                                                                               \
    if (OMNI_RING_SETPRIO) __builtin_amdgcn_s_setprio(1);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    OMNI_RING_PAIR(buf, 0)                                                                                   \
    if (PREFETCH) af[buf][0] = lds_read16<0, ABL == 3>(aa_);                                                           \
    if (!OMNI_RING_STAGGER || !wm) { DMA_A; }                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    OMNI_RING_PAIR(buf, 1)                                                                                   \
    if (PREFETCH) af[buf][1] = lds_read16<32 * 64, ABL == 3>(aa_);                                                     \
    if (OMNI_RING_STAGGER && wm) { DMA_A; }                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    OMNI_RING_PAIR(buf, 2)                                                                                   \
    if (PREFETCH) af[buf][2] = lds_read16<2 * 32 * 64, ABL == 3>(aa_);                                                 \
    if (!OMNI_RING_STAGGER || !wm) { DMA_B; }                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    OMNI_RING_PAIR(buf, 3)                                                                                   \
    if (PREFETCH) {                                                                                          \
      af[buf][3] = lds_read16<3 * 32 * 64, ABL == 3>(aa_);                                                             \
      wf[buf][0] = lds_read16<ROP_BYTES, ABL == 3>(wa_);                                                               \
      wf[buf][1] = lds_read16<ROP_BYTES + 32 * 64, ABL == 3>(wa_);                                                     \
    }                                                                                                        \
    if (OMNI_RING_STAGGER && wm) { DMA_B; }                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if (OMNI_RING_SETPRIO) __builtin_amdgcn_s_setprio(0);                                                                           \
  } while (0)
  // B_0
  if (LEAD - 1 < nst) {
    asm volatile(OMNI_RING_VMCNT ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (LEAD < nst) issue_stage(LEAD, LEAD);
  OMNI_RING_READ(0, 0, 0);
  OMNI_RING_READ(1, 0, 1);
  // DMA for stage st+LEAD goes into the slot of stage st-1 (free since B_st) and is issued ONE PIECE AT A TIME
  // between the MFMAs of iteration st: a global_load_lds costs its wave 60-185 issue cycles, which the partner
  // wave on the SIMD covers with its own MFMAs; a burst of 4-8 right after the barrier stalls all waves at once.
  int slot = 0;                       // slot of stage st
  int pslot = RSTAGES - 1;            // slot of stage st-1 (stage LEAD was issued into slot LEAD at B_0)
  for (int st = 0; st + 1 < nst; ++st) {               // every stage but the last: stage st+1 exists
    const int nslot = (slot + 1 == RSTAGES) ? 0 : slot + 1;
    const bool dma = st > 0 && st + LEAD < nst;
    const int dst = st + LEAD;
    // k-step 2*st   (prefetches stage st+1 / k-step 0: visible since B_st)
    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    OMNI_RING_MMA(0, true, nslot, 0, if (dma && ABL != 1) issue_piece(pslot, dst, 0), if (dma && ABL != 1) issue_piece(pslot, dst, 1));
    // k-step 2*st+1 (prefetches stage st+1 / k-step 1)
    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    OMNI_RING_MMA(1, true, nslot, 1, if (dma && ABL != 1) issue_piece(pslot, dst, 2), if (dma && ABL != 1) issue_piece(pslot, dst, 3));
    // ---- B_{st+1}: own pieces of stages <= st+2 landed; afterwards the slot of stage st is free
    if (ABL != 4) {
      if (st + LEAD < nst && ABL != 1) {
        asm volatile(OMNI_RING_VMCNT ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("" ::: "memory");
    pslot = slot;
    slot = nslot;
  }
  // last stage: nothing left to prefetch
  asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
  OMNI_RING_MMA(0, false, 0, 0, (void)0, (void)0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  OMNI_RING_MMA(1, false, 0, 1, (void)0, (void)0);
#undef OMNI_RING_READ
#undef OMNI_RING_ADDR
#undef OMNI_RING_MMA
#undef OMNI_RING_PAIR

  if (COALESCED) gemm_epilogue_lds<EPI>(P, G, acc, m0, n0, wm, wn, l31, hi, smem, tid);
  else gemm_epilogue<EPI>(P, G, acc, m0, n0, wm, wn, l31, hi);
  if (bid + (int)gridDim.x < nwg) __syncthreads();   // the next tile's DMA reuses the LDS the epilogue has just read
  }  // tile loop
}


// ------------------------------------------------------------------------------------------------
// Ping-pong variant (OMNI_GEMM_VARIANT=3): BK = 64 K-tiles, the two wave groups of the workgroup (wm = 0 / 1: the two
// waves that share each SIMD) run HALF A PHASE APART, so that on every SIMD one wave is inside an MFMA-only cluster while
// its partner issues its fragment reads and LDS-DMA pieces (cdna_hip_programming.md "256^2 8-phase template", T3+T4+T5).
// In the ring kernel all 8 waves execute the same instruction mix in lock step: whenever the L1's request queue is full
// every wave stalls on its next global_load_lds and the matrix pipe drains (MFMA busy 62 %, TCP_PENDING_STALL 47 %).
//
// A K-tile is consumed as four QUADRANTS of the wave's 128 x 64 output (A rows mq*64.. x W rows nq*32..), one quadrant
// (8 MFMAs = 256 matrix-pipe cycles) per phase, in the order (mq,nq) = (0,0) (0,1) (1,1) (1,0); the operands of a K-tile
// are staged as four HALF-TILES of 128 rows x 64 k (16 KiB each), in the order the phases first need them:
//   h0 = A rows of mq 0 (both wm)   h1 = W rows of nq 0 (all wn)   h2 = W rows of nq 1   h3 = A rows of mq 1
// phase 0 reads h0 + h1 (12 ds_read_b128), phase 1 reads h2 (4), phase 2 reads h3 (8, into the registers of h0), phase 3
// reads nothing: 24 reads per K-tile (the minimum), every half-tile is read in exactly one phase.
// LDS = ring of 8 half-tile slots (2 K-tiles, 128 KiB); half-tile j = 4*tile + h lives in slot j % 8 and is issued SIX
// phases ahead (in phase g = j - 6, two 1-KiB pieces per wave).  Barrier slots: group 0 runs [load g | B | mma g | B],
// group 1 the same one barrier later.
//   RAW: a wave ends every load section with vmcnt(8): its pieces of half-tiles <= g + 2 have landed; the partner group's
//        covering wait is at most one barrier older than the first read (phase g + 1 reads half-tiles <= g + 2).
//   WAR: the slot of half-tile j was last read in phase <= j - 8 by both groups (their lgkmcnt(0) sits behind the barrier
//        that follows the read); group 0 re-issues it in phase j - 6, two full phases later.
// LDS image of a slot: [128 rows][64 k] bf16 (128-B rows), 16-B chunk index XOR ((row >> 1) & 7) on the DMA source and
// on the ds_read_b128 address (conflict-free for the four 16-lane groups).  A DMA piece = 8 rows x 128 B: whole cache
// lines in the row-major layout, 2 x (8 rows x 64 B contiguous) in the K32-blocked layouts.
// Same accumulator layout and the same k order as the ring kernel: results are bit-identical to it.
// ------------------------------------------------------------------------------------------------
// counted wait at the end of a load section: the youngest FOUR half-tiles (x 2 pieces per wave) may stay in flight across a barrier
#define OMNI_PP_VMCNT "s_waitcnt vmcnt(8)"
#define OMNI_PP_VMCNT_LOOP OMNI_PP_VMCNT
// (No s_setprio around the clusters since round 4: with the restructured K-loop the same box measured 0 .. +5 % without it on all
// four DiT shapes, bit-identical — the load sections carry no VALU work the clusters would have to outrank.  What else was
// measured on this loop and is NOT in the kernel any more — DMA pieces inside the clusters, a second A-fragment register set
// (reads 4 / 4 / 8 / 8 per phase), the closing barrier signalled early, 32-MFMA clusters on two big phases per K-tile, the
// 32x32x16 MFMA shape, and the timing ablations — is in DESIGN.md 7 items 14-18, 28-30 with the logs under profiles/.)
// The cluster's MFMAs are issued from inline asm: as builtins they are "pure" nodes that hipcc's instruction selection
// is free to sink below s_setprio 0 / the closing s_barrier and interleave with the NEXT phase's reads (observed) —
// which destroys exactly the phase separation this kernel is about.  Volatile asm keeps program order with respect to the
// barriers, waits and reads.  Hazards the compiler can no longer see: the operands come from ds_reads retired by the
// explicit lgkmcnt(0); consecutive MFMAs alternate between two accumulators, SrcC == vDst exactly (the interlocked case);
// the epilogue's first VALU read of an accumulator is kept >= 18 wait states away by the s_nops behind the k-loop.
// MFMA shape of the ping-pong kernel: v_mfma_f32_16x16x32_bf16 — the same flops per matrix-pipe cycle as 32x32x16, but
// an instruction carries 32 k instead of 16: half the accumulator read-modify-write traffic per flop.  On this power-limited
// part that is clock: swapping only the instruction shape (same loads, same barriers) measured 1127 -> 1194 TF/s (+6 %,
// profiles/r02_gemm_mfma_shape.log).  Both shapes add their products to the fp32 accumulator in 8-k groups in ascending k:
// the results are bit-identical to the 32x32x16 build and to the ring kernel (tests/test_gpu_ops.py asserts it).
OMNI_DEVINL void pp_mfma16(f32x4_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// fp8 (omni_gemm_params.fp8): ONE v_mfma_scale_f32_16x16x128_f8f6f4 replaces the two 16x16x32 bf16 MFMAs of a (block, block) pair.
// Its A / B operand is 8 VGPRs = 32 fp8 of one row: bytes 0-15 = k 16g .. 16g+15, bytes 16-31 = k 64+16g .. 64+16g+15 (g = lane >> 4;
// measured with .gpu_scratch-style probes, profiles/r03_mx_mfma_operand_probe.log) — exactly the two 16-B chunks (g, 4 + g) of a
// 128-B LDS row that the bf16 kernel's ks = 0 / 1 fragment reads already fetch.  So the fp8 kernel IS the bf16 kernel on a matrix
// with half the columns (byte-identical DMA, LDS image, swizzle and fragment reads), with this instruction in the cluster.
// Block scales (E8M0, one byte per lane and 32-k block) are the constant 1.0 (0x7f): the per-row / per-channel fp32 scales are
// applied to the accumulators in the epilogue.
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8_t;
OMNI_DEVINL void pp_mfma_fp8(f32x4_t& acc, const bf16x8_t& a_lo, const bf16x8_t& a_hi, const bf16x8_t& b_lo, const bf16x8_t& b_hi,
                             uint32_t one) {
  const u32x4_t al = __builtin_bit_cast(u32x4_t, a_lo), ah = __builtin_bit_cast(u32x4_t, a_hi);
  const u32x4_t bl = __builtin_bit_cast(u32x4_t, b_lo), bh = __builtin_bit_cast(u32x4_t, b_hi);
  const u32x8_t a = {al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
  const u32x8_t b = {bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(one));
}
// Dev-only phase probe (-DOMNI_DEV -DOMNI_PP_PROBE=1, dev/gemm_pp_probe.h; tools/probe/pp_probe.cpp reads it back): s_memtime
// stamps at five points of every phase.  Product builds: the OMNI_PP_STAMP / OMNI_PP_PROBE_ACCUM hooks below are empty and the
// device code is byte-identical to a build without them.
#ifndef OMNI_PP_PROBE
#define OMNI_PP_PROBE 0
#endif
#if OMNI_PP_PROBE
#include "dev/gemm_pp_probe.h"
#else
#define OMNI_PP_STAMP(v) ((void)0)
#define OMNI_PP_PROBE_ACCUM(nq, mq) ((void)0)
#endif
// The whole-launch split-K instance (SPLITK == 1) runs the steady-state K-loop too when its piece has at least this many K-tiles
// (round 6: bit-identical; same box, 60-layer forwards: one 384^2 request -2.7 %, one 512^2 -2.2 %, two / four 256^2 requests
// -1.9 %; the 8-K-tile pieces of one 256^2 CFG pair's out-projection gain nothing — profiles/r06b_ab_splitk_steady_loop.log)
#ifndef OMNI_SPLITK_STEADY_MIN_KT
#define OMNI_SPLITK_STEADY_MIN_KT 12
#endif
constexpr int PBK = 64;
constexpr int PSLOT_BYTES = 128 * PBK * 2;    // 16 KiB per half-tile
constexpr int PLDS_BYTES = 8 * PSLOT_BYTES;   // 128 KiB ring

// NSPLIT > 1 (host: small grids, see launch()): split-K.  Workgroup b computes K-tiles [split * nkt, (split+1) * nkt) of
// tile b / nsplit with split = b % nsplit and stores its fp32 accumulators to P.splitk_ws[split][row][col]; the epilogue runs
// in gemm_splitk_finish_kernel.  A DiT forward over one or two 256x256 images has 36 workgroups in its N = 3072 GEMMs, each
// streaming its 256 x K weight panel at the pace of one CU's k-loop (~1.2 us per K-tile): the weights arrive at ~1 TB/s.
// SPLITK == 2 (host: tail_split_factor()): TAIL split.  A launch of more than one round of workgroups whose LAST round is thin —
// 1548 tiles on 256 CUs are 6 rounds + 12 tiles, the N = 3072 GEMMs of one 2048^2 request — runs its first `tail_first` tiles
// exactly as the unsplit kernel does (same code path, same bits) and the remaining tiles `nsplit` ways along K: blocks
// tail_first + (tile - tail_first) * nsplit + split, dispatched last, fp32 partials in a tile-compact workspace
// splitk_ws[tile - tail_first][split][256][256], epilogue in gemm_tail_finish_kernel.  The thin round then costs ~1 / nsplit of a
// tile time instead of a whole one.
// SPLITK == 3 (round 6): whole-launch split-K as SPLITK == 1, reduced INSIDE the launch (no finish kernel, no second pass of
// the partials through a launch boundary).  Every workgroup of the grid is resident at once (host: grid <= CUs, one workgroup
// per CU), so the nsplit workgroups of a tile can wait for each other: each stores its fp32 partial WRITE-THROUGH (sc1: the
// bytes leave the XCD's L2 at once, no release fence — cdna_hip_programming.md 6 Guideline 16 "publish-large"), drains its stores,
// takes a ticket on the tile's arrival counter (relaxed, agent scope), polls it (one lane, relaxed) until all nsplit
// tickets are drawn, takes ONE agent-scope acquire and then runs the usual epilogue from the partials over ITS share of the
// tile's rows — a reduce-scatter: nobody reads more than one tile's worth of partials, and the sum is formed in split order by
// the same code as the finish kernel's (bit-identical results).  The nsplit workgroups of a tile are placed on ONE XCD (block
// b runs on XCD b % 8): b -> (xcd = b & 7, j = b >> 3), tile = (j / nsplit) * 8 + xcd, split = j % nsplit; the grid is padded to
// 8 * ceil(tiles / 8) * nsplit blocks and the pad blocks return at once.  Counters: two int32 per tile (arrivals, departures) at
// the end of the workspace, zero before the launch and zero after it (the last workgroup to leave a tile resets both).
constexpr int SPLITK_CNT_INTS = 512;              // [0,128) arrivals, [128,256) departures, [256] sticky "a wait timed out" flag
constexpr int SPLITK_FBATCH = 2;                  // rows per thread and pipeline stage of the in-launch finish: 8 batches of 32 rows per tile
template <int EPI, int SPLITK = 0, int FP8 = 0>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_pp_kernel(const omni_gemm_params P, int mtiles0, int tiles_m,
                                                                     int tiles_n, int GROUP_M, int nsplit, int tail_first) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = tiles_m * tiles_n;
  constexpr bool WHOLE = SPLITK == 1 || SPLITK == 3;                         // every tile of the launch is split
  const bool tail = SPLITK == 2 && (int)blockIdx.x >= tail_first;          // uniform over the workgroup
  const int split = SPLITK == 1 ? (int)blockIdx.x % nsplit
                    : SPLITK == 3 ? ((int)blockIdx.x >> 3) % nsplit
                                  : (tail ? ((int)blockIdx.x - tail_first) % nsplit : 0);
  const int bid = SPLITK == 1 ? (int)blockIdx.x / nsplit
                  : SPLITK == 3 ? (((int)blockIdx.x >> 3) / nsplit) * 8 + ((int)blockIdx.x & 7)
                                : (tail ? tail_first + ((int)blockIdx.x - tail_first) / nsplit : (int)blockIdx.x);
  if (SPLITK == 3 && bid >= nwg) return;                                   // a pad block of the XCD-aligned grid
  const bool partial = WHOLE || tail;                                      // this workgroup leaves an fp32 partial tile
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int band_sz = GROUP_M * tiles_n;
  const int band = lid / band_sz, in_band = lid - band * band_sz;
  const int first_m = band * GROUP_M;
  const int gm = min(GROUP_M, tiles_m - first_m);
  const int mt = first_m + in_band % gm;
  const int nt = in_band / gm;
  const int gi = (mt >= mtiles0) ? 1 : 0;
  const omni_gemm_group G = pick_group(P, gi);
  const int m0 = (gi ? mt - mtiles0 : mt) * BM;
  const int n0 = nt * BN;
  const int M = G.M, N = P.N, K = P.K;
  if (G.tile_skip && G.tile_skip[gi ? mt - mtiles0 : mt]) return;   // device-side predicate (omni_teacache)
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5;

  // ---- per-lane DMA source byte offsets (32-bit, SADDR form): [mq | nq][piece i] --------------------------------
  uint32_t a_off[2][2], w_off[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = (wave * 2 + i) * 8 + (lane >> 3);          // row inside the half-tile slot (0..127)
    const int c = (lane & 7) ^ ((lr >> 1) & 7);               // logical 16-B chunk landing in physical chunk lane & 7
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int ar = min(m0 + (lr >> 6) * 128 + q * 64 + (lr & 63), M - 1);
      if (G.a_row_map) ar = G.a_row_map[ar];
      const int64_t ae = G.a_k32_rows ? ((int64_t)(c >> 2) * G.a_k32_rows + ar) * 32 + (c & 3) * 8
                                      : (int64_t)ar * G.lda + c * 8;
      a_off[q][i] = (uint32_t)(ae * 2);
      const int wr = min(n0 + (lr >> 5) * 64 + q * 32 + (lr & 31), N - 1);
      const int64_t we = P.w_k32_blocked ? ((int64_t)(c >> 2) * N + wr) * 32 + (c & 3) * 8 : (int64_t)wr * K + c * 8;
      w_off[q][i] = (uint32_t)(we * 2);
    }
  }
  // bytes between two K-tiles of one operand: two [rows][32] slabs in the K32-blocked layouts, 128 B in row-major
  const int64_t astep = G.a_k32_rows ? (int64_t)G.a_k32_rows * 128 : 128;
  const int64_t wstep = P.w_k32_blocked ? (int64_t)N * 128 : 128;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // K-tiles of this workgroup: all of them; an nsplit-th (whole-launch split-K: nsplit divides the count); tail split: the
  // split-th of nsplit near-equal contiguous ranges
  const int nkt_all = K / PBK;
  const int kt0 = WHOLE ? split * (nkt_all / nsplit) : (tail ? (int)((long)split * nkt_all / nsplit) : 0);
  const int nkt = WHOLE ? nkt_all / nsplit : (tail ? (int)((long)(split + 1) * nkt_all / nsplit) - kt0 : nkt_all);
  const char* const Ab = reinterpret_cast<const char*>(G.A) + (SPLITK ? (int64_t)kt0 * astep : 0);
  const char* const Wb = reinterpret_cast<const char*>(G.W) + (SPLITK ? (int64_t)kt0 * wstep : 0);
  // piece i (0 / 1) of half-tile h (compile time) of K-tile `tile`
#define OMNI_PP_ISSUE_PIECE(h, tile, i)                                                                     \
  do {                                                                                                      \
    const int t_ = (tile);                                                                                  \
    const uint32_t dst_ = lds0 + (((t_ & 1) * 4 + (h)) * PSLOT_BYTES) + (wave * 2 + (i)) * 1024;              \
    if ((h) == 0 || (h) == 3) glds16_saddr(Ab + t_ * astep, a_off[(h) == 3][i], dst_);                      \
    else glds16_saddr(Wb + t_ * wstep, w_off[(h) == 2][i], dst_);                                           \
  } while (0)
#define OMNI_PP_ISSUE(h, tile)                                                                              \
  do {                                                                                                      \
    OMNI_PP_ISSUE_PIECE(h, tile, 0);                                                                        \
    OMNI_PP_ISSUE_PIECE(h, tile, 1);                                                                        \
  } while (0)

  // ---- per-lane fragment read offsets (16x16x32: lane = row l15 of a 16-row block, k = 32*ks + 8*g .. +8):
  //      row * 128 + ((ks*4 + g) ^ swz) * 16, swz = (row >> 1) & 7 = (l15 >> 1) & 7; 16-row blocks via the offset immediate.
  //      Every 16-lane group reads 16 consecutive rows at one logical chunk: (row & 1) * 8 + (chunk ^ swz) covers all 16
  //      16-B bank groups -> conflict-free, like the 32x32 pattern.
  const int l15 = lane & 15, g4 = lane >> 4;
  uint32_t a_rd[2], w_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const uint32_t chunk = ((uint32_t)(ks * 4 + g4) ^ ((l15 >> 1) & 7)) << 4;
    a_rd[ks] = lds0 + (wm * 64 + l15) * 128 + chunk;
    w_rd[ks] = lds0 + (wn * 32 + l15) * 128 + chunk;
  }

  // ---- prologue: half-tiles 0..5 in flight; 0 and 1 landed before the first read ---------------------------------
  OMNI_PP_ISSUE(0, 0); OMNI_PP_ISSUE(1, 0); OMNI_PP_ISSUE(2, 0); OMNI_PP_ISSUE(3, 0);
  if (nkt > 1) { OMNI_PP_ISSUE(0, 1); OMNI_PP_ISSUE(1, 1); }
  // accumulators start at the bias.  The bias loads sit BEHIND the prologue's DMA issue: hipcc retires them with vmcnt(0)
  // (it cannot see the asm DMAs), which placed between the DMA issues would drain the first pieces before the rest is sent.
  f32x4_t acc[4][8];                               // [16-column block][16-row block]: C[mb*16 + l15][nb*16 + 4*g4 + j]
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = n0 + wn * 64 + nb * 16 + g4 * 4;
    u32x2_t b = {0u, 0u};
    if (!partial && !FP8 && G.bias && n < N) b = *reinterpret_cast<const u32x2_t*>(G.bias + n);   // split-K: the finish adds the bias; fp8: the epilogue does (after the scales)
    const f32x4_t bini = {bf16_lo(b[0]), bf16_hi(b[0]), bf16_lo(b[1]), bf16_hi(b[1])};
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[nb][mb] = bini;
  }
  // tail split: everything the partial store needs, formed HERE as per-lane values (a pointer pair and two small counts in
  // VGPRs) so that bid / split / nsplit / tail_first are dead across the K-loop — kept as scalars they pushed the QKV instance
  // into SGPR spills inside the loop
  float* tail_wsp = nullptr;
  int tail_rows = 0, tail_cols = 0;
  if (SPLITK == 2 && tail) {
    const int rl = wm * 128 + l15, cl = wn * 64 + g4 * 4;
    tail_wsp = P.splitk_ws + ((int64_t)(bid - tail_first) * nsplit + split) * (BM * BN) + rl * BN + cl;
    tail_rows = M - m0 - rl;                       // block mb exists iff tail_rows > 16 mb
    tail_cols = N - n0 - cl;                       // block nb exists iff tail_cols > 16 nb
    asm volatile("" : "+v"(tail_wsp), "+v"(tail_rows), "+v"(tail_cols));
  }
  if (nkt > 1) {
    asm volatile(OMNI_PP_VMCNT ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (wm) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0
  asm volatile("" ::: "memory");

  // wf[nq][16-row block of the 32 W rows][ks]; afx[16-row block of the 64 A rows][ks] (mq 0 and mq 1 share the registers)
  bf16x8_t wf[2][2][2], afx[4][2];
  bf16x8_t (&afy)[4][2] = afx;
  uint32_t mx_one = 0x7f7f7f7fu;                   // E8M0 1.0 in every byte: the MFMA's block scales (fp8 build only)
  asm volatile("" : "+v"(mx_one));
#define OMNI_PP_READ_A(AF, sb)                                                             \
  do {                                                                                     \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) {                                  \
      AF[0][ks_] = lds_read16<0>(a_rd[ks_] + (sb));                                        \
      AF[1][ks_] = lds_read16<16 * 128>(a_rd[ks_] + (sb));                                 \
      AF[2][ks_] = lds_read16<32 * 128>(a_rd[ks_] + (sb));                                 \
      AF[3][ks_] = lds_read16<48 * 128>(a_rd[ks_] + (sb));                                 \
    }                                                                                      \
  } while (0)
#define OMNI_PP_READ_W(nq, sb)                                                             \
  do {                                                                                     \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) {                                  \
      wf[nq][0][ks_] = lds_read16<0>(w_rd[ks_] + (sb));                                    \
      wf[nq][1][ks_] = lds_read16<16 * 128>(w_rd[ks_] + (sb));                             \
    }                                                                                      \
  } while (0)
// the quadrant's 16 MFMAs: 8 accumulators round-robin, each touched again 8 instructions (128 pipe cycles) later; bf16: pair p =
// (ks, mb) = (p / 4, p % 4); fp8: ONE scaled MFMA per (block, block) pair covers both ks
#define OMNI_PP_CLUSTER(nq, mq, AF)                                                                        \
  if (FP8) {                                                                                               \
    _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_) {                                                  \
      pp_mfma_fp8(acc[2 * (nq)][4 * (mq) + mb_], wf[nq][0][0], wf[nq][0][1], AF[mb_][0], AF[mb_][1], mx_one);     \
      pp_mfma_fp8(acc[2 * (nq) + 1][4 * (mq) + mb_], wf[nq][1][0], wf[nq][1][1], AF[mb_][0], AF[mb_][1], mx_one); \
    }                                                                                                      \
  } else                                                                                                   \
  _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) {                                                       \
      pp_mfma16(acc[2 * (nq)][4 * (mq) + (p_ & 3)], wf[nq][0][p_ >> 2], AF[p_ & 3][p_ >> 2]);              \
      pp_mfma16(acc[2 * (nq) + 1][4 * (mq) + (p_ & 3)], wf[nq][1][p_ >> 2], AF[p_ & 3][p_ >> 2]);          \
    }
// end of a load section: counted DMA wait, barrier, fragments arrived; then the MFMA cluster and the second barrier.
// `landed_ok`: the counted wait is valid (enough younger pieces were issued behind the ones that must have landed).
#define OMNI_PP_MMA(nq, mq, AF, landed_ok)                                                                 \
  do {                                                                                                     \
    if (landed_ok) asm volatile(OMNI_PP_VMCNT_LOOP ::: "memory");                                          \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
    OMNI_PP_STAMP(pb_t2);                                                                                  \
    __builtin_amdgcn_s_barrier();                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    OMNI_PP_STAMP(pb_t3);                                                                                  \
    OMNI_PP_CLUSTER(nq, mq, AF)                                                                            \
    OMNI_PP_PROBE_ACCUM(nq, mq); /* T1..T3 of this phase, T4 / T5 of the previous one: all landed long ago */ \
    OMNI_PP_STAMP(pb_t4);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    OMNI_PP_STAMP(pb_t5);                                                                                  \
    asm volatile("" ::: "memory");                                                                         \
  } while (0)
// one phase of the GENERAL loop (run-time `do_issue`): both DMA pieces of half-tile (h, tile) in the load section
#define OMNI_PP_PHASE(nq, mq, AF, do_issue, h, tile)                                                       \
  do {                                                                                                     \
    if (do_issue) OMNI_PP_ISSUE(h, tile);                                                                  \
    OMNI_PP_STAMP(pb_t1);                                                                                  \
    OMNI_PP_MMA(nq, mq, AF, do_issue);                                                                     \
  } while (0)

#if OMNI_PP_PROBE
  OMNI_PP_PROBE_DECLS();
#endif
  int t_first = 0;
  // STEADY STATE (every K-tile whose successors t + 1 and t + 2 both exist; split-K pieces of >= 12 K-tiles): everything the scalar unit
  // decides per phase in the general loop below is compile-time here — the loop runs two K-tiles per trip (ring parity = a
  // constant), `t + 1 < nkt` / `t + 2 < nkt` hold by construction, one running pointer per operand instead of `base + t * step`
  // per phase, M0 = one s_add of a per-wave constant and an immediate: 0.25 scalar instructions per MFMA (1.17 in the general
  // loop; the vendor kernel: 0.33), load section 246 -> 154 cycles per phase by the phase probe.  The fp8
  // instance runs the same loop since round 5 (its clusters are half as long: the scalar stream weighed twice as much).
  if (SPLITK != 1 || nkt >= OMNI_SPLITK_STEADY_MIN_KT) {
    const uint32_t lds_w = lds0 + (uint32_t)(wave * 2048);           // this wave's two pieces inside a half-tile slot
    const char* a_nx = Ab + astep;                                  // K-tile t + 1 of either operand (t = 0)
    const char* w_nx = Wb + wstep;
#define OMNI_PP_ISSUE_C(h, base, par)                                                                      \
  do {                                                                                                     \
    constexpr uint32_t so_ = (uint32_t)(((par) * 4 + (h)) * PSLOT_BYTES);                                  \
    const uint32_t v0_ = ((h) == 0 || (h) == 3) ? a_off[(h) == 3][0] : w_off[(h) == 2][0];                 \
    const uint32_t v1_ = ((h) == 0 || (h) == 3) ? a_off[(h) == 3][1] : w_off[(h) == 2][1];                 \
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"                      \
                 :: "s"(lds_w), "i"(so_), "v"(v0_), "s"(base) : "memory", "scc");                        \
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"                      \
                 :: "s"(lds_w), "i"(so_ + 1024u), "v"(v1_), "s"(base) : "memory", "scc");                \
  } while (0)
    // one K-tile of ring parity PAR
    auto ktile = [&](auto par_c) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par_c)::value;
      constexpr uint32_t sb = (uint32_t)(PAR * 4 * PSLOT_BYTES);
      // phase 0: quadrant (mq 0, nq 0); DMA: half-tile 2 (W rows of nq 1) of K-tile t + 1
      OMNI_PP_READ_A(afx, sb);
      OMNI_PP_READ_W(0, sb + PSLOT_BYTES);
      OMNI_PP_ISSUE_C(2, w_nx, PAR ^ 1);
      OMNI_PP_STAMP(pb_t1);
      OMNI_PP_MMA(0, 0, afx, true);
      // phase 1: quadrant (mq 0, nq 1); DMA: half-tile 3 (A rows of mq 1) of K-tile t + 1
      OMNI_PP_READ_W(1, sb + 2 * PSLOT_BYTES);
      OMNI_PP_ISSUE_C(3, a_nx, PAR ^ 1);
      OMNI_PP_STAMP(pb_t1);
      OMNI_PP_MMA(1, 0, afx, true);
      a_nx += astep;                                                // K-tile t + 2
      w_nx += wstep;
      // phase 2: quadrant (mq 1, nq 1); DMA: half-tile 0 (A rows of mq 0) of K-tile t + 2
      OMNI_PP_READ_A(afy, sb + 3 * PSLOT_BYTES);
      OMNI_PP_ISSUE_C(0, a_nx, PAR);
      OMNI_PP_STAMP(pb_t1);
      OMNI_PP_MMA(1, 1, afy, true);
      // phase 3: quadrant (mq 1, nq 0); DMA: half-tile 1 (W rows of nq 0) of K-tile t + 2
      OMNI_PP_ISSUE_C(1, w_nx, PAR);
      OMNI_PP_STAMP(pb_t1);
      OMNI_PP_MMA(0, 1, afy, true);
    };
#pragma unroll 1
    for (; t_first + 3 < nkt; t_first += 2) {
#if OMNI_PP_PROBE
      pb_snap = (t_first | 1) == ((nkt >> 1) | 1);
#endif
      ktile(std::integral_constant<int, 0>{});
      ktile(std::integral_constant<int, 1>{});
    }
#undef OMNI_PP_ISSUE_C
  }
  // GENERAL loop: the last two or three K-tiles (their successors may not exist), short K, and the split-K instance
#pragma unroll 1
  for (int t = t_first; t < nkt; ++t) {
#if OMNI_PP_PROBE
    pb_snap = t == (nkt >> 1);                   // the middle K-tile: T3 of its four phases, T4 of its phases 0-2 (slot 3: the T4 before it)
#endif
    const uint32_t sb = (uint32_t)((t & 1) * 4 * PSLOT_BYTES);
    const bool n1 = t + 1 < nkt, n2 = t + 2 < nkt;
    // phase 0: quadrant (mq 0, nq 0)
    OMNI_PP_READ_A(afx, sb);
    OMNI_PP_READ_W(0, sb + PSLOT_BYTES);
    OMNI_PP_PHASE(0, 0, afx, n1, 2, t + 1);
    // phase 1: quadrant (mq 0, nq 1)
    OMNI_PP_READ_W(1, sb + 2 * PSLOT_BYTES);
    OMNI_PP_PHASE(1, 0, afx, n1, 3, t + 1);
    // phase 2: quadrant (mq 1, nq 1)
    OMNI_PP_READ_A(afy, sb + 3 * PSLOT_BYTES);
    OMNI_PP_PHASE(1, 1, afy, n2, 0, t + 2);
    // phase 3: quadrant (mq 1, nq 0)
    OMNI_PP_PHASE(0, 1, afy, n2, 1, t + 2);
  }
  if (!wm) __builtin_amdgcn_s_barrier();         // group 0 waits for group 1's last cluster
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // asm MFMA results -> first compiler-visible VALU read
  // (the OMNI_PP_* loop macros stay defined for the persistent variant below, which repeats this loop; #undef'd behind it)
#if OMNI_PP_PROBE
  OMNI_PP_PROBE_WRITE();
#endif

  if (SPLITK == 2 && tail) {
    // tail split: tile-compact fp32 partial [256][256] of (tile, split); rows / columns past M / N are never read back
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      if (tail_rows <= mb * 16) continue;             // rows of this lane's 16-row blocks that exist (m0 + rl < M)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        if (tail_cols > nb * 16) *reinterpret_cast<f32x4_t*>(tail_wsp + mb * 16 * BN + nb * 16) = acc[nb][mb];
    }
    return;
  }
  if constexpr (SPLITK == 3) {
    const int64_t mtot = P.g[0].M + (P.ngroups > 1 ? P.g[1].M : 0);
    float* const wsp = P.splitk_ws + ((int64_t)split * mtot + (gi ? P.g[0].M : 0)) * N;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      const int row = m0 + wm * 128 + mb * 16 + l15;
      if (row >= M) continue;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int col = n0 + wn * 64 + nb * 16 + g4 * 4;
        float* dst = wsp + (int64_t)row * N + col;
        if (col < N) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(acc[nb][mb]) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // EVERY storing wave drains its write-through stores
    __syncthreads();
    int* const cnt = reinterpret_cast<int*>(P.splitk_ws + P.splitk_ws_floats) - SPLITK_CNT_INTS;
    if (tid == 0) __hip_atomic_fetch_add(cnt + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // this workgroup's share of the tile's LIVE 32-row batches (a text tile may hold fewer than 256 rows)
    constexpr int RS3 = NTHREADS / 32 * SPLITK_FBATCH;          // 32 rows per batch
    const int live = (min(BM, M - m0) + RS3 - 1) / RS3;
    const int b0 = split * live / nsplit, b1 = (split + 1) * live / nsplit;
    if (b1 > b0) {
      if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(cnt + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nsplit) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1 << 22)) {                            // ~ a second: never on a healthy launch; do not hang the device
            __hip_atomic_store(cnt + 256, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE acquire (drops this CU's stale L1 lines), then plain loads
      }
      __syncthreads();
      const EpiFromPartials src = {P.splitk_ws, mtot * N, gi ? (int64_t)P.g[0].M : 0, nsplit, b0, b1, N, 0, 0};
      auto nothing = []() {};
      gemm_epilogue_lds_impl<EPI, decltype(nothing), EpiFromPartials, NTHREADS, SPLITK_FBATCH>(P, G, m0, n0, nullptr, tid, nothing, src);
    }
    if (tid == 0) {                                             // (behind this workgroup's own wait, if it had one)
      const int d = __hip_atomic_fetch_add(cnt + 128 + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == nsplit - 1) {                                    // the last to leave: everyone is past the poll
        __hip_atomic_store(cnt + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cnt + 128 + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  if (SPLITK == 1) {
    // fp32 partial tile: lane (l15, g4) holds C[mb*16 + l15][nb*16 + 4*g4 .. +4]: 16-B stores, 64 contiguous bytes per row
    float* const wsp = P.splitk_ws + ((int64_t)split * (P.g[0].M + (P.ngroups > 1 ? P.g[1].M : 0)) + (gi ? P.g[0].M : 0)) * N;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      const int row = m0 + wm * 128 + mb * 16 + l15;
      if (row >= M) continue;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int col = n0 + wn * 64 + nb * 16 + g4 * 4;
        if (col < N) *reinterpret_cast<f32x4_t*>(wsp + (int64_t)row * N + col) = acc[nb][mb];
      }
    }
    return;
  }
  // K32-blocked outputs are stored straight from the accumulators
  if constexpr (EPI == OMNI_EPI_BIAS || EPI == OMNI_EPI_BIAS_GELU_TANH) {
    if (G.out_k32_rows && !G.out_row_map) {          // uniform over the workgroup
      gemm_epilogue_direct_k32<EPI, FP8>(P, G, acc, m0, n0, wm, wn, l15, g4);
      return;
    }
  }
  if (FP8) gemm_epilogue_lds_fp8<EPI>(P, G, acc, m0, n0, wm, wn, l15, g4, smem, tid);
  else gemm_epilogue_lds<EPI>(P, G, acc, m0, n0, wm, wn, l15, g4, smem, tid);
}


#if OMNI_PP_PROBE   // the probe instruments the one-shot kernel only: the persistent variant below expands the same loop macros
#undef OMNI_PP_STAMP
#undef OMNI_PP_PROBE_ACCUM
#define OMNI_PP_STAMP(v) ((void)0)
#define OMNI_PP_PROBE_ACCUM(nq, mq) ((void)0)
#endif

#ifdef OMNI_DEV
#include "dev/gemm_family8_persistent_pingpong.inc"
#endif
#undef OMNI_PP_PHASE
#undef OMNI_PP_MMA
#undef OMNI_PP_CLUSTER
#undef OMNI_PP_READ_W
#undef OMNI_PP_READ_A
#undef OMNI_PP_ISSUE
#undef OMNI_PP_ISSUE_PIECE

// Split-K finish: one workgroup per output tile (the same workgroup -> tile map as the ping-pong kernel), phase 2 of the
// row-coalesced epilogue with C taken from the fp32 partials.
// Round 6: EIGHT workgroups of 256 threads per tile (32 rows each) instead of four of 512: a 36-tile launch (one 256^2 CFG pair)
// finishes on 288 workgroups instead of 144 — the finish is a latency-bound stream of fp32 partials and the chip has 256 CUs.
// Round 6, third session: ONE row per thread — 32 workgroups of 8 rows per tile — and the partial loads of a row in front of its
// map-dependent loads: 60-layer forwards, same box, -0.7 % (two 256^2 requests), -1.3 % (one 384^2 request, config 1) against four
// rows per thread (profiles/r06c_ab_finish_variants.log); same bits (the order of the adds is the split order either way).
#ifndef OMNI_FIN_BATCH
#define OMNI_FIN_BATCH 1          // rows per thread of a finish workgroup (dev A/B: 1, 2, 4)
#endif
constexpr int FIN_THREADS = 256, FIN_BATCH = OMNI_FIN_BATCH, FIN_PER_TILE = BM / (FIN_THREADS / 32 * FIN_BATCH);   // 8 rows in flight per pass x FIN_BATCH per thread = the workgroup's rows
template <int EPI>
__global__ __launch_bounds__(FIN_THREADS) void gemm_splitk_finish_kernel(const omni_gemm_params P, int mtiles0, int tiles_m,
                                                                       int tiles_n, int GROUP_M, int nsplit) {
  const int bid = blockIdx.x / FIN_PER_TILE, quarter = blockIdx.x % FIN_PER_TILE;      // (`quarter`: the tile's 32-row slice)
  const int nwg = tiles_m * tiles_n;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int band_sz = GROUP_M * tiles_n;
  const int band = lid / band_sz, in_band = lid - band * band_sz;
  const int first_m = band * GROUP_M;
  const int gm = min(GROUP_M, tiles_m - first_m);
  const int mt = first_m + in_band % gm;
  const int nt = in_band / gm;
  const int gi = (mt >= mtiles0) ? 1 : 0;
  const omni_gemm_group G = pick_group(P, gi);
  const int m0 = (gi ? mt - mtiles0 : mt) * BM;
  const int n0 = nt * BN;
  if (G.tile_skip && G.tile_skip[gi ? mt - mtiles0 : mt]) return;
  if (m0 + quarter * (BM / FIN_PER_TILE) >= G.M) return;
  const int64_t mtot = P.g[0].M + (P.ngroups > 1 ? P.g[1].M : 0);
  const EpiFromPartials src = {P.splitk_ws, mtot * P.N, gi ? (int64_t)P.g[0].M : 0, nsplit, quarter, quarter + 1, P.N, 0, 0};
  auto nothing = []() {};
  gemm_epilogue_lds_impl<EPI, decltype(nothing), EpiFromPartials, FIN_THREADS, FIN_BATCH>(P, G, m0, n0, nullptr, (int)threadIdx.x, nothing, src);
}

// Tail-split finish: the same epilogue over the tail tiles only (four workgroups of 64 rows per tile), C = the sum of the tile's
// nsplit compact partials in split order.
template <int EPI>
__global__ __launch_bounds__(FIN_THREADS) void gemm_tail_finish_kernel(const omni_gemm_params P, int mtiles0, int tiles_m,
                                                                     int tiles_n, int GROUP_M, int nsplit, int tail_first) {
  const int tl = blockIdx.x / FIN_PER_TILE, quarter = blockIdx.x % FIN_PER_TILE;
  const int bid = tail_first + tl;
  const int nwg = tiles_m * tiles_n;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int band_sz = GROUP_M * tiles_n;
  const int band = lid / band_sz, in_band = lid - band * band_sz;
  const int first_m = band * GROUP_M;
  const int gm = min(GROUP_M, tiles_m - first_m);
  const int mt = first_m + in_band % gm;
  const int nt = in_band / gm;
  const int gi = (mt >= mtiles0) ? 1 : 0;
  const omni_gemm_group G = pick_group(P, gi);
  const int m0 = (gi ? mt - mtiles0 : mt) * BM;
  const int n0 = nt * BN;
  if (G.tile_skip && G.tile_skip[gi ? mt - mtiles0 : mt]) return;
  if (m0 + quarter * (BM / FIN_PER_TILE) >= G.M) return;
  const EpiFromPartials src = {P.splitk_ws + (int64_t)tl * nsplit * (BM * BN), BM * BN, 0, nsplit, quarter, quarter + 1, BN, m0, n0};
  auto nothing = []() {};
  gemm_epilogue_lds_impl<EPI, decltype(nothing), EpiFromPartials, FIN_THREADS, FIN_BATCH>(P, G, m0, n0, nullptr, (int)threadIdx.x, nothing, src);
}

#ifdef OMNI_DEV
#include "dev/gemm_family5_6_two_phase_pingpong_and_2_w4.inc"
#endif

#ifdef OMNI_DEV
#include "dev/gemm_family4_q4_four_waves_agpr.inc"
#endif

#ifdef OMNI_DEV
#include "dev/gemm_family7_v4_vendor_schedule.inc"
#endif

int gemm_group_m() {
  // dev knob: OMNI_GEMM_GROUP_M = row-tiles per L2 band
  static int v = -1;
  if (v < 0) {
    v = omni_dev_env_int("OMNI_GEMM_GROUP_M", GROUP_M_DEFAULT);
    if (v < 1) v = 1;
  }
  return v;
}

// Kernel family of a call: 3 = ping-pong (default), 1 = ring (omni_gemm_params.kernel_hint == OMNI_GEMM_KERNEL_RING, and the
// fallback for what the ping-pong kernel does not take).  -DOMNI_DEV builds: kernel_hint >= 16 selects family hint - 16
// (0 first design, 2 hipcc-scheduled 4-wave, 4 Q4, 5 / 6 two-phase ping-pong), else OMNI_GEMM_VARIANT from the environment.
int gemm_variant(const omni_gemm_params* p) {
  if (p->kernel_hint == OMNI_GEMM_KERNEL_RING) return 1;
#ifdef OMNI_DEV
  if (p->kernel_hint >= 16) return p->kernel_hint - 16;
  static const int env = omni_dev_env_int("OMNI_GEMM_VARIANT", 3);
  return env;
#else
  return 3;
#endif
}

bool gemm_variant_blocked_ok(const omni_gemm_params* p) { const int v = gemm_variant(p); return v == 1 || v >= 3; }

// The row-coalesced epilogue moves 16 B per thread: every output / residual / gate pointer and stride must allow it.
// (OMNI_GEMM_EPI_LDS=0 forces the direct epilogue: dev knob.)
bool epilogue_rows_coalescable(const omni_gemm_params* p) {
  static int knob = -1;
  if (knob < 0) {
    knob = omni_dev_env_int("OMNI_GEMM_EPI_LDS", 1);
  }
  if (!knob) return false;
  static_assert(EPI_LDS_BYTES <= RLDS_BYTES, "C tile must fit the operand ring's LDS");
  for (int g = 0; g < p->ngroups; ++g) {
    const omni_gemm_group& G = p->g[g];
    if (!omni_aligned16(G.out) || (!G.out_k32_rows && (G.ldo % 8) != 0)) return false;
    if (G.bias && !omni_aligned16(G.bias)) return false;
    if (p->epilogue == OMNI_EPI_BIAS_GATE_RES &&
        (!omni_aligned16(G.res) || !omni_aligned16(G.gate) || (G.ldres % 8) != 0 || (G.gate_item_stride % 8) != 0))
      return false;
    if ((p->epilogue == OMNI_EPI_BIAS_SPLIT3 || p->epilogue == OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE) &&
        (!omni_aligned16(G.out1) || !omni_aligned16(G.out2)))
      return false;
  }
  return true;
}

int gemm_num_cus() { return omni_num_cus(); }
// SADDR-form DMA needs every byte offset of both operands to fit 32 bits (dev knob OMNI_GEMM_SADDR=0 disables it)
bool ring_saddr_ok(const omni_gemm_params* p) {
  static int knob = -1;
  if (knob < 0) {
    knob = omni_dev_env_int("OMNI_GEMM_SADDR", 1);
  }
  if (!knob) return false;
  const int64_t lim = 1ll << 32;
  if (p->w_k32_blocked ? ((int64_t)p->N * RBK * 2 >= lim) : ((int64_t)p->N * p->K * 2 >= lim)) return false;
  for (int g = 0; g < p->ngroups; ++g) {
    const omni_gemm_group& G = p->g[g];
    if (G.a_k32_rows) {
      if ((int64_t)G.a_k32_rows * RBK * 2 >= lim) return false;
    } else if (G.a_row_map || (int64_t)G.M * G.lda * 2 >= lim) {
      return false;
    }
  }
  return true;
}
bool gemm_persistent() {
  // dev knob: OMNI_GEMM_PERSISTENT=1 -> one workgroup per CU walking the tiles.  Measured neutral (0.3688 vs 0.3690 images/s,
  // same box): workgroup launch is not what the ~8 us per-tile overhead consists of, so the default stays one workgroup
  // per tile (hardware-ordered dispatch).
  static int v = -1;
  if (v < 0) {
    v = omni_dev_env_int("OMNI_GEMM_PERSISTENT", 0);
  }
  return v != 0;
}

#ifdef OMNI_DEV
#include "dev/gemm_dev_persistent_predicate.inc"
#endif

// Split-K factor for a launch of the ping-pong kernel, 1 = off.  On when the caller gave a workspace and the grid would leave
// at least half of the chip idle (<= 128 tiles in <= 10 row tiles: a forward over one or two small images): the
// largest s in {8, 6, 4, 3, 2} that divides the K-tile count, keeps tiles * s within one round of the CUs and fits the
// workspace.  s depends on the TILE COUNTS only, so a batch and its sequence-parallel shards (fewer rows, same N and K)
// take the same decision whenever both stay under the limits.
int splitk_factor(const omni_gemm_params* p, int tiles_m, int tiles_n) {
  static int knob = -1;                       // dev knob: OMNI_GEMM_SPLITK=0 disables
  if (knob < 0) {
    knob = omni_dev_env_int("OMNI_GEMM_SPLITK", 1);
  }
  if (!knob || !p->splitk_ws || p->splitk_ws_floats <= 0) return 1;
  const int tiles = tiles_m * tiles_n;
  if (tiles > 128 || (tiles_m > 10 && p->kernel_hint != OMNI_GEMM_KERNEL_SPLITK_TALL) || (p->N % 4) != 0 ||
      (reinterpret_cast<uintptr_t>(p->splitk_ws) & 15))
    return 1;
  const int nkt = p->K / PBK;
  const int64_t mtot = p->g[0].M + (p->ngroups > 1 ? p->g[1].M : 0);
  const int cus = gemm_num_cus();
  const int min_kt = omni_dev_env_int("OMNI_GEMM_SPLITK_MIN_KT", 4);      // dev knob (-DOMNI_DEV builds only): K-tiles per piece
  for (int s : {8, 6, 4, 3, 2}) {
    if (nkt % s == 0 && nkt / s >= min_kt && tiles * s <= cus && (int64_t)s * mtot * p->N <= p->splitk_ws_floats) return s;
  }
  return 1;
}

// Tail split of a launch of `tiles` 256x256 tiles (see the kernel comment): returns the split factor of the tiles of its last,
// partial round (1 = off) and *tail_first = the number of tiles that run unsplit.  The rule is MEASURED, not modelled:
//   * only a THIN tail is split (at most a quarter round of tiles): 12 K-tiles per piece at K = 3072 (factor 4), 24 at K = 12288
//     (factor 8).  Whole 60-layer forwards, same box (profiles/r06_ab_splits_*.log): two 1024^2 requests (780 / 2340 / 3120
//     tiles = 3 / 9 / 12 rounds + 12 / 36 / 48) -4.4 %, one 2048^2 request (6 / 18 / 24 rounds + 12 / 36 / 48) -1.0 %;
//   * a BIG tail (140 .. 236 tiles: one, three or five 1024^2 requests) is left alone.  Launched alone, every factor >= 4 gains
//     1 .. 12 % there too (profiles/r06_tail_split_sweep.log) — the 35 .. 60 MB of fp32 partials stay in the 256 MB Infinity Cache —
//     but inside a forward the same launches cost +2.5 .. +8 % of the step: 150 .. 250 MB of partials per GEMM evict the
//     activations the next kernel reads.  (Factors 2 and 3 on a big tail are slower even alone: two nearly synchronous
//     sub-rounds of long pieces.)
// Like splitk_factor() it looks at tile counts only; unlike it, it changes the fp32 summation order of SOME tiles of a launch, so
// results agree with the unsplit kernel to that order on the tail tiles and bit for bit elsewhere (tests/test_gpu_tail_split.py).
int tail_split_factor(const omni_gemm_params* p, int tiles, int* tail_first) {
  static const int knob = omni_dev_env_int("OMNI_GEMM_TAILSPLIT", 1);     // dev knob (-DOMNI_DEV builds only)
  if (!knob || !p->splitk_ws || p->splitk_ws_floats <= 0 || p->kernel_hint == OMNI_GEMM_KERNEL_NO_TAIL_SPLIT) return 1;
  if ((reinterpret_cast<uintptr_t>(p->splitk_ws) & 15) || (p->N % 4) != 0) return 1;
  const int cus = gemm_num_cus();
  const int tt = tiles % cus;
  if (tiles <= cus || tt == 0) return 1;
  const int nkt = p->K / PBK;
  *tail_first = tiles - tt;
  const int force = omni_dev_env_int("OMNI_GEMM_TAIL_NS", 0);            // dev knob, read per call: a fixed factor (sweeps)
  if (force > 0) {
    const long blocks = (long)tt * force;
    return (force == 1 || nkt / force < 2 || blocks * (long)(BM * BN) > p->splitk_ws_floats) ? 1 : force;
  }
  if (tt * 4 > cus) return 1;                                            // a big tail: see above
  int ns = nkt >= 96 ? 8 : 4;
  while (ns > 1 && (nkt / ns < 8 || (long)tt * ns * (long)(BM * BN) > p->splitk_ws_floats)) ns >>= 1;
  return ns;
}

// In-launch reduce of a whole-launch split-K (gemm_bf16_pp_kernel SPLITK == 3): OPT-IN (omni_gemm_params.kernel_hint =
// OMNI_GEMM_KERNEL_SPLITK_IN_LAUNCH).  Needs the XCD-aligned grid within one round of the CUs (all of a tile's workgroups resident
// together), at most 128 tiles and room for the counters behind the partials; otherwise the two-kernel path runs.
// MEASURED, same box, whole 60-layer forwards (profiles/r06b_ab_inlaunch_by_factor.log, r06b_ab_smallm.log): it is NOT the
// default because it does not pay where the split is deep — one 256^2 CFG pair (36 tiles x 6: +2.5 .. +3.8 % per forward,
// config 1 +3 %): a workgroup exchanges partials with its peers at the per-block rate of the fabric (~65 GB/s, 256 .. 393 KB
// each way), which takes as long as the 288-workgroup finish kernel it replaces, and the early finishers idle.  Where the
// split is 2-way and the grid nearly fills the chip it gains 1.1 .. 1.3 % (one 512^2 request, four 256^2 requests: 120 tiles x
// 2); at 60 tiles x 2 (two 256^2 requests) it loses 3.5 % (120 workgroups finishing 128 rows each against 480 finish
// workgroups).
bool splitk_in_launch_ok(const omni_gemm_params* p, int tiles, int nsplit) {
  if (p->kernel_hint != OMNI_GEMM_KERNEL_SPLITK_IN_LAUNCH &&
      omni_dev_env_int("OMNI_GEMM_SPLITK_INLAUNCH", 0) < nsplit)           // dev knob (-DOMNI_DEV builds only): largest nsplit taken
    return false;
  if (tiles > 128) return false;
  const int64_t mtot = p->g[0].M + (p->ngroups > 1 ? p->g[1].M : 0);
  if ((int64_t)nsplit * mtot * p->N + SPLITK_CNT_INTS > p->splitk_ws_floats) return false;
  return ((tiles + 7) / 8) * 8 * nsplit <= gemm_num_cus();
}

template <int EPI>
int launch(const omni_gemm_params* p, hipStream_t s) {
  const int mt0 = (p->g[0].M + BM - 1) / BM;
  const int mt1 = p->ngroups > 1 ? (p->g[1].M + BM - 1) / BM : 0;
  const int tiles_m = mt0 + mt1, tiles_n = (p->N + BN - 1) / BN;
  static std::atomic<uint64_t> attr_done{0};     // per device and per epilogue instance (common.h omni_once_per_device)
  auto set_attrs = []() -> bool {
#ifdef OMNI_DEV
#include "dev/gemm_dev_launch_attrs.inc"
#endif
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_ring_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, RLDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_ring_kernel<EPI, 0, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, RLDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_ring_kernel<EPI, 0, true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, RLDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, RLDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp_kernel<EPI, 1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, RLDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp_kernel<EPI, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, RLDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp_kernel<EPI, 3>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, RLDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp_kernel<EPI, 0, 1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, RLDS_BYTES) != hipSuccess)
      return false;
#ifdef OMNI_DEV
#include "dev/gemm_dev_launch_attr_persistent.inc"
#endif
    return true;
  };
  OMNI_TRY_STATUS(omni_once_per_device(attr_done, set_attrs));
  if (p->fp8) {
    if (p->kernel_hint == OMNI_GEMM_KERNEL_SPLITK_DEFER_FINISH) return OMNI_ERR_UNSUPPORTED;       // fp8: no split-K
    // the fp8 operands are, byte for byte, K32-blocked bf16 matrices with K / 2 columns: the kernel runs on that view
    omni_gemm_params q = *p;
    q.K = p->K / 2;
    if (!epilogue_rows_coalescable(&q) || !ring_saddr_ok(&q)) return OMNI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, 0, 1>), dim3(tiles_m * tiles_n), dim3(NTHREADS), RLDS_BYTES, s, q, mt0, tiles_m,
                       tiles_n, gemm_group_m(), 1, 0);
    OMNI_CHECK_LAUNCH();
    return OMNI_OK;
  }
#ifdef OMNI_DEV
#include "dev/gemm_dev_launch_first_w4_pp2.inc"
#endif
#ifdef OMNI_DEV
#include "dev/gemm_dev_launch_q4_v4.inc"
#endif
#ifdef OMNI_DEV
#include "dev/gemm_dev_launch_persistent.inc"
#endif
  if (false) {
  }
  else if (gemm_variant(p) >= 3 && p->K % PBK == 0 && epilogue_rows_coalescable(p) && ring_saddr_ok(p)) {
    const int nsplit = splitk_factor(p, tiles_m, tiles_n);
    int tail_first = 0;
    const int tail_ns = nsplit > 1 ? 1 : tail_split_factor(p, tiles_m * tiles_n, &tail_first);
    if (p->kernel_hint == OMNI_GEMM_KERNEL_SPLITK_DEFER_FINISH) {
      // ABI v13: the K-split main kernel only; the caller's consumer (omni_splitk_finish_adaln_pair) reduces the partials
      if (nsplit <= 1) return OMNI_ERR_UNSUPPORTED;
      hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, 1>), dim3(tiles_m * tiles_n * nsplit), dim3(NTHREADS), RLDS_BYTES, s, *p,
                         mt0, tiles_m, tiles_n, gemm_group_m(), nsplit, 0);
    } else if (nsplit > 1 && splitk_in_launch_ok(p, tiles_m * tiles_n, nsplit)) {
      // counters: zeroed on the stream before the launch (a memset node under capture), left zero by the launch
      int* cnt = reinterpret_cast<int*>(p->splitk_ws + p->splitk_ws_floats) - SPLITK_CNT_INTS;
      if (hipMemsetAsync(cnt, 0, SPLITK_CNT_INTS * sizeof(int), s) != hipSuccess) return OMNI_ERR_LAUNCH;
      hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, 3>), dim3(((tiles_m * tiles_n + 7) / 8) * 8 * nsplit), dim3(NTHREADS),
                         RLDS_BYTES, s, *p, mt0, tiles_m, tiles_n, gemm_group_m(), nsplit, 0);
    } else if (nsplit > 1) {
      hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, 1>), dim3(tiles_m * tiles_n * nsplit), dim3(NTHREADS), RLDS_BYTES, s, *p,
                         mt0, tiles_m, tiles_n, gemm_group_m(), nsplit, 0);
      OMNI_CHECK_LAUNCH();
      hipLaunchKernelGGL((gemm_splitk_finish_kernel<EPI>), dim3(tiles_m * tiles_n * FIN_PER_TILE), dim3(FIN_THREADS), 0, s, *p, mt0,
                         tiles_m, tiles_n, gemm_group_m(), nsplit);
    } else if (tail_ns > 1) {
      const int tt = tiles_m * tiles_n - tail_first;
      hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, 2>), dim3(tail_first + tt * tail_ns), dim3(NTHREADS), RLDS_BYTES, s, *p,
                         mt0, tiles_m, tiles_n, gemm_group_m(), tail_ns, tail_first);
      OMNI_CHECK_LAUNCH();
      hipLaunchKernelGGL((gemm_tail_finish_kernel<EPI>), dim3(tt * FIN_PER_TILE), dim3(FIN_THREADS), 0, s, *p, mt0, tiles_m, tiles_n,
                         gemm_group_m(), tail_ns, tail_first);
    } else {
      hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI>), dim3(tiles_m * tiles_n), dim3(NTHREADS), RLDS_BYTES, s, *p, mt0, tiles_m,
                         tiles_n, gemm_group_m(), 1, 0);
    }
  }
  else if (p->kernel_hint == OMNI_GEMM_KERNEL_SPLITK_DEFER_FINISH) return OMNI_ERR_UNSUPPORTED;   // no split-K outside the ping-pong kernel
  else if (epilogue_rows_coalescable(p)) {
    int grid = tiles_m * tiles_n;
    if (gemm_persistent() && grid > gemm_num_cus()) grid = gemm_num_cus() & ~7;
    if (ring_saddr_ok(p))
      hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI, 0, true, true>), dim3(grid), dim3(NTHREADS), RLDS_BYTES, s, *p, mt0,
                         tiles_m, tiles_n, gemm_group_m());
    else
      hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI, 0, true>), dim3(grid), dim3(NTHREADS), RLDS_BYTES, s, *p, mt0,
                         tiles_m, tiles_n, gemm_group_m());
  }
  else
    hipLaunchKernelGGL(gemm_bf16_ring_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(NTHREADS), RLDS_BYTES, s, *p, mt0,
                       tiles_m, tiles_n, gemm_group_m());
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

}  // namespace



#ifdef OMNI_DEV
#include "dev/gemm_dev_entry_points.inc"
#endif

namespace {
int gemm_validate(const omni_gemm_params* p) {
  if (!p || p->ngroups < 1 || p->ngroups > 2 || p->N <= 0 || p->K <= 0) return OMNI_ERR_BAD_ARG;
  if (p->K % RBK != 0 || p->N % 8 != 0) return OMNI_ERR_UNSUPPORTED;
  if (p->fp8 != 0 && p->fp8 != 1) return OMNI_ERR_BAD_ARG;
  if (p->fp8) {                                      // e4m3 operands: K64-blocked layouts only, whole 128-k tiles, fp32 scales
    if (p->K % 128 != 0 || p->N % 4 != 0 || !p->w_k32_blocked) return OMNI_ERR_UNSUPPORTED;
    for (int g = 0; g < p->ngroups; ++g) {
      if (!p->g[g].a_scale || !p->g[g].w_scale) return OMNI_ERR_BAD_ARG;
      if (!p->g[g].a_k32_rows) return OMNI_ERR_UNSUPPORTED;
      if (reinterpret_cast<uintptr_t>(p->g[g].w_scale) & 15) return OMNI_ERR_ALIGN;
    }
  }
  for (int g = 0; g < p->ngroups; ++g) {
    const omni_gemm_group& G = p->g[g];
    if (!G.A || !G.W || !G.out || G.M <= 0) return OMNI_ERR_BAD_ARG;
    if (!omni_aligned16(G.A) || !omni_aligned16(G.W) || (!G.a_k32_rows && (G.lda % 8) != 0)) return OMNI_ERR_ALIGN;
    if ((reinterpret_cast<uintptr_t>(G.out) & 7) || (!G.out_k32_rows && (G.ldo % 4) != 0)) return OMNI_ERR_ALIGN;
    if (p->epilogue == OMNI_EPI_BIAS_GATE_RES) {
      if (!G.res || !G.gate || (!G.row_item_map && G.rows_per_item <= 0)) return OMNI_ERR_BAD_ARG;
      if ((G.ldres % 4) != 0 || (G.gate_item_stride % 4) != 0) return OMNI_ERR_ALIGN;
    }
    if (p->epilogue == OMNI_EPI_BIAS_SPLIT3 || p->epilogue == OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE) {
      if (!G.out1 || !G.out2) return OMNI_ERR_BAD_ARG;
    }
    if (p->epilogue == OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE) {
      if (!G.qk_norm_q_w || !G.qk_norm_k_w || !G.qk_rope_cos || !G.qk_rope_sin || !G.qk_row_pos) return OMNI_ERR_BAD_ARG;
      if (!omni_aligned16(G.qk_norm_q_w) || !omni_aligned16(G.qk_norm_k_w) ||
          (reinterpret_cast<uintptr_t>(G.qk_rope_cos) & 7) || (reinterpret_cast<uintptr_t>(G.qk_rope_sin) & 7))
        return OMNI_ERR_ALIGN;
    }
  }
  const bool split3 = p->epilogue == OMNI_EPI_BIAS_SPLIT3 || p->epilogue == OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE;
  if (split3 && (p->split_n <= 0 || p->split_n % 32 != 0 || p->N != 3 * p->split_n)) return OMNI_ERR_UNSUPPORTED;
  if (p->epilogue == OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE) {     // the fused norm+RoPE exists in the ring kernel's coalesced epilogue only
    if (p->split_n % 128 != 0 || !gemm_variant_blocked_ok(p)) return OMNI_ERR_UNSUPPORTED;
    if (!epilogue_rows_coalescable(p)) return OMNI_ERR_ALIGN;
  }
  for (int g = 0; g < p->ngroups; ++g) {
    const omni_gemm_group& G = p->g[g];
    if (G.a_k32_rows < 0 || G.out_k32_rows < 0) return OMNI_ERR_BAD_ARG;
    if (G.a_k32_rows && !G.a_row_map && G.a_k32_rows < G.M) return OMNI_ERR_BAD_ARG;
    if (G.out_k32_rows) {
      if (p->epilogue != OMNI_EPI_BIAS && p->epilogue != OMNI_EPI_BIAS_GELU_TANH) return OMNI_ERR_UNSUPPORTED;
      if (p->N % 32 != 0 || (!G.out_row_map && G.out_k32_rows < G.M)) return OMNI_ERR_BAD_ARG;
    }
    if ((G.a_k32_rows || G.out_k32_rows) && !gemm_variant_blocked_ok(p)) return OMNI_ERR_UNSUPPORTED;
    if (G.out_k32_rows && !epilogue_rows_coalescable(p)) return OMNI_ERR_ALIGN;
  }
  if (p->w_k32_blocked != 0 && p->w_k32_blocked != 1) return OMNI_ERR_BAD_ARG;
  if (p->w_k32_blocked && !gemm_variant_blocked_ok(p)) return OMNI_ERR_UNSUPPORTED;   // only the ring / ping-pong kernels read that layout
  return OMNI_OK;
}
}  // namespace

// ABI v13: the whole-launch split-K factor the call would use (launch()'s rule, evaluated without launching)
extern "C" int omni_gemm_splitk_factor(const omni_gemm_params* p) {
  if (gemm_validate(p) != OMNI_OK) return 0;
  if (p->fp8) return 1;
  if (!(gemm_variant(p) >= 3 && p->K % PBK == 0 && epilogue_rows_coalescable(p) && ring_saddr_ok(p))) return 1;
  const int mt0 = (p->g[0].M + BM - 1) / BM;
  const int mt1 = p->ngroups > 1 ? (p->g[1].M + BM - 1) / BM : 0;
  return splitk_factor(p, mt0 + mt1, (p->N + BN - 1) / BN);
}

extern "C" int omni_gemm_bf16(const omni_gemm_params* p, omni_stream stream) {
  OMNI_TRY_STATUS(gemm_validate(p));
  // a deferred finish exists only for a call that splits: refused here, before anything touches the device
  if (p->kernel_hint == OMNI_GEMM_KERNEL_SPLITK_DEFER_FINISH && omni_gemm_splitk_factor(p) <= 1) return OMNI_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (p->epilogue) {
    case OMNI_EPI_BIAS: return launch<OMNI_EPI_BIAS>(p, s);
    case OMNI_EPI_BIAS_GELU_TANH: return launch<OMNI_EPI_BIAS_GELU_TANH>(p, s);
    case OMNI_EPI_BIAS_GATE_RES: return launch<OMNI_EPI_BIAS_GATE_RES>(p, s);
    case OMNI_EPI_BIAS_SPLIT3: return launch<OMNI_EPI_BIAS_SPLIT3>(p, s);
    case OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE: return launch<OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE>(p, s);
    default: return OMNI_ERR_BAD_ARG;
  }
}
