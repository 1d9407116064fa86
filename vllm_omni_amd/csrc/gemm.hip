// Grouped bf16 GEMM with fused epilogues for gfx950:  Y_g = epi(A_g · W_gᵀ + bias_g).
//
// Geometry (one workgroup = one 256x256 output tile, 8 waves = 2(M) x 4(N), 512 threads, 1 WG / CU):
//   BK = 64;  LDS = 2 stages x (A tile 32 KiB + W tile 32 KiB) = 128 KiB of the CU's 160 KiB.
//   Staging is direct-to-LDS (global_load_lds_dwordx4, 1 KiB per wave-instruction = 8 rows x 128 B,
//   i.e. full 128-B lines of 8 consecutive rows).  The LDS image is row-major [256][64] bf16 whose
//   16-byte chunk index is XOR-swizzled with (row>>1)&7; because the DMA destination is lane-linear
//   the swizzle is applied to the per-lane SOURCE address and again on the ds_read_b128 address
//   (cdna_hip_programming.md rule 21).  With it the four 16-lane groups of a ds_read_b128 hit 16
//   distinct 16-byte slots (conflict-free).
//   MFMA: v_mfma_f32_32x32x16_bf16 with SWAPPED operands (A-operand = W fragment, B-operand = A fragment),
//   so a wave's accumulator holds Cᵀ blocks: lane (l&31) owns ONE output row and, per 4 registers,
//   FOUR CONSECUTIVE output columns -> bias/gate/residual are 8-byte vector loads and the store is
//   8 bytes per lane (hi/lo half-waves adjacent -> 16 B contiguous per row).
//   One barrier per K tile: wait own DMA (vmcnt 0) -> barrier -> issue DMA for tile t+1 into the other
//   stage -> 24 ds_read_b128 + 32 MFMA on tile t.
//   blockIdx is remapped so that each XCD (block b runs on XCD b % 8) owns a contiguous band of tiles
//   and walks it GROUP_M row-tiles at a time: neighbours share A/W panels in that XCD's private L2.
//
// Roofline: MFMA-bound.  Algorithmic work = 2*M*N*K flop per launch.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int NTHREADS = 512;
constexpr int TILE_BYTES = BM * BK * 2;      // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;   // 128 KiB
constexpr int GROUP_M = 4;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

OMNI_DEVINL void glds16(const void* gsrc, uint32_t lds_byte_addr) {
  // wave-uniform LDS base (goes to M0); the hardware adds lane*16.
  __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)(uintptr_t)lds_byte_addr, 16, 0, 0);
}

template <int EPI>
OMNI_DEVINL void gemm_epilogue(const omni_gemm_params& P, const omni_gemm_group& G, f32x16_t (&acc)[2][4], int m0,
                               int n0, int wm, int wn, int l31, int hi) {
  const int M = G.M, N = P.N;
  // ---- epilogue: acc[nb][mb][4q+j] = C[m][n],  m = m0+wm*128+mb*32+l31,  n = n0+wn*64+nb*32+8q+4hi+j
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int m = m0 + wm * 128 + mb * 32 + l31;
    if (m >= M) continue;
    const int64_t orow = G.out_row_map ? G.out_row_map[m] : m;
    int item = 0;
    if (EPI == OMNI_EPI_BIAS_GATE_RES) item = G.row_item_map ? G.row_item_map[m] : m / G.rows_per_item;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + nb * 32 + q * 8 + hi * 4;
        if (n >= N) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[nb][mb][q * 4 + j];
        if (G.bias) {
          const u32x2_t b = *reinterpret_cast<const u32x2_t*>(G.bias + n);
          v[0] += bf16_lo(b[0]); v[1] += bf16_hi(b[0]); v[2] += bf16_lo(b[1]); v[3] += bf16_hi(b[1]);
        }
        uint16_t* dst;
        if (EPI == OMNI_EPI_BIAS_SPLIT3) {
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
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_kernel(const omni_gemm_params P, int mtiles0, int tiles_m,
                                                                  int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- tile id: XCD-aware bijective remap, then GROUP_M banding -------------------------------
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int band_sz = GROUP_M * tiles_n;
  const int band = lid / band_sz, in_band = lid - band * band_sz;
  const int first_m = band * GROUP_M;
  const int gm = min(GROUP_M, tiles_m - first_m);
  const int mt = first_m + in_band % gm;
  const int nt = in_band / gm;
  const int gi = (mt >= mtiles0) ? 1 : 0;
  const omni_gemm_group& G = P.g[gi];
  const int m0 = (gi ? mt - mtiles0 : mt) * BM;
  const int n0 = nt * BN;
  const int M = G.M, N = P.N, K = P.K;

  // ---- per-lane DMA source pointers: 4 A rows + 4 W rows, fixed over the K loop ---------------
  const uint16_t* a_src[4];
  const uint16_t* w_src[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 8 + (lane >> 3);            // tile row this lane feeds
    const int c = (lane & 7) ^ ((r >> 1) & 7);                 // logical k-chunk landing in phys chunk lane&7
    int ar = min(m0 + r, M - 1);
    if (G.a_row_map) ar = G.a_row_map[ar];
    a_src[j] = G.A + (int64_t)ar * G.lda + c * 8;
    const int wr = min(n0 + r, N - 1);
    w_src[j] = G.W + (int64_t)wr * K + c * 8;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;  // LDS byte address of the dynamic region
  auto issue_stage = [&](int stage, int kt) {
    const uint32_t base = lds0 + stage * STAGE_BYTES + (wave * 4) * 1024;
    const int koff = kt * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(a_src[j] + koff, base + j * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(w_src[j] + koff, base + TILE_BYTES + j * 1024);
  };

  // ---- per-lane fragment read offsets ---------------------------------------------------------
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5;
  uint32_t a_row_off[4], a_swz[4], w_row_off[2], w_swz[2];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int r = wm * 128 + mb * 32 + l31;
    a_row_off[mb] = r * 128;
    a_swz[mb] = (r >> 1) & 7;
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int r = wn * 64 + nb * 32 + l31;
    w_row_off[nb] = TILE_BYTES + r * 128;
    w_swz[nb] = (r >> 1) & 7;
  }

  f32x16_t acc[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][mb][i] = 0.0f;

  const int nkt = K / BK;
  issue_stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nkt) issue_stage(cur ^ 1, kt + 1);
    const char* sb = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint32_t ch = ks * 2 + hi;
      bf16x8_t wf[2], af[4];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        wf[nb] = *reinterpret_cast<const bf16x8_t*>(sb + w_row_off[nb] + ((ch ^ w_swz[nb]) << 4));
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        af[mb] = *reinterpret_cast<const bf16x8_t*>(sb + a_row_off[mb] + ((ch ^ a_swz[mb]) << 4));
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], af[mb], acc[nb][mb], 0, 0, 0);
    }
  }

  gemm_epilogue<EPI>(P, G, acc, m0, n0, wm, wn, l31, hi);
}

// ------------------------------------------------------------------------------------------------
// Ring variant: BK = 32 per stage, 5-deep LDS ring (5 x 32 KiB = the CU's whole 160 KiB), DMA issued FOUR
// stages (= 2 BK64 tiles, ~2 us) ahead and retired with a COUNTED vmcnt(12); raw s_barrier (a __syncthreads()
// would drain the in-flight LDS-DMA with vmcnt(0)).  Motivation (profiles/r01_rocprof_summary_v1.txt): with the
// 2-stage BK=64 pipeline the MFMA pipe is busy only ~50 % of resident wave time because one tile of compute
// (2048 cycles/SIMD) is shorter than the time a 64 KiB tile needs to arrive through the 64 B/clk load path.
// LDS image per operand per stage: row-major [256][32] bf16 (64-B rows), 16-B chunk index XOR (row>>2)&3.
// One DMA wave-instruction = 16 rows x 64 B.
// ------------------------------------------------------------------------------------------------
constexpr int RBK = 32, RSTAGES = 5;
constexpr int ROP_BYTES = BM * RBK * 2;        // 16 KiB per operand per stage
constexpr int RSTAGE_BYTES = 2 * ROP_BYTES;    // 32 KiB
constexpr int RLDS_BYTES = RSTAGES * RSTAGE_BYTES;  // 160 KiB

template <int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_ring_kernel(const omni_gemm_params P, int mtiles0,
                                                                       int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int band_sz = GROUP_M * tiles_n;
  const int band = lid / band_sz, in_band = lid - band * band_sz;
  const int first_m = band * GROUP_M;
  const int gm = min(GROUP_M, tiles_m - first_m);
  const int mt = first_m + in_band % gm;
  const int nt = in_band / gm;
  const int gi = (mt >= mtiles0) ? 1 : 0;
  const omni_gemm_group& G = P.g[gi];
  const int m0 = (gi ? mt - mtiles0 : mt) * BM;
  const int n0 = nt * BN;
  const int M = G.M, N = P.N, K = P.K;

  const uint16_t* a_src[2];
  const uint16_t* w_src[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    int ar = min(m0 + r, M - 1);
    if (G.a_row_map) ar = G.a_row_map[ar];
    a_src[j] = G.A + (int64_t)ar * G.lda + c * 8;
    const int wr = min(n0 + r, N - 1);
    w_src[j] = G.W + (int64_t)wr * K + c * 8;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  auto issue_stage = [&](int slot, int st) {
    const uint32_t base = lds0 + slot * RSTAGE_BYTES + (wave * 2) * 1024;
    const int koff = st * RBK;
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(a_src[j] + koff, base + j * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(w_src[j] + koff, base + ROP_BYTES + j * 1024);
  };

  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5;
  uint32_t a_row_off[4], a_swz[4], w_row_off[2], w_swz[2];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int r = wm * 128 + mb * 32 + l31;
    a_row_off[mb] = r * 64;
    a_swz[mb] = (r >> 2) & 3;
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int r = wn * 64 + nb * 32 + l31;
    w_row_off[nb] = ROP_BYTES + r * 64;
    w_swz[nb] = (r >> 2) & 3;
  }

  f32x16_t acc[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][mb][i] = 0.0f;

  const int nst = K / RBK;
#pragma unroll
  for (int s = 0; s < RSTAGES - 1; ++s)
    if (s < nst) issue_stage(s, s);
  int slot = 0, islot = RSTAGES - 1;
  for (int s = 0; s < nst; ++s) {
    if (s + (RSTAGES - 2) < nst) {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // stages s+1..s+3 (3 x 4 DMAs per wave) may stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + (RSTAGES - 1) < nst) issue_stage(islot, s + (RSTAGES - 1));
    const char* sb = smem + slot * RSTAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint32_t ch = ks * 2 + hi;
      bf16x8_t wf[2], af[4];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        wf[nb] = *reinterpret_cast<const bf16x8_t*>(sb + w_row_off[nb] + ((ch ^ w_swz[nb]) << 4));
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        af[mb] = *reinterpret_cast<const bf16x8_t*>(sb + a_row_off[mb] + ((ch ^ a_swz[mb]) << 4));
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], af[mb], acc[nb][mb], 0, 0, 0);
    }
    slot = (slot + 1 == RSTAGES) ? 0 : slot + 1;
    islot = (islot + 1 == RSTAGES) ? 0 : islot + 1;
  }
  gemm_epilogue<EPI>(P, G, acc, m0, n0, wm, wn, l31, hi);
}


int gemm_variant() {
  // dev knob: OMNI_GEMM_VARIANT=0 -> 2-stage BK=64 pipeline, 1 (default) -> 5-stage BK=32 ring
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("OMNI_GEMM_VARIANT");
    v = e ? atoi(e) : 1;
  }
  return v;
}

template <int EPI>
int launch(const omni_gemm_params* p, hipStream_t s) {
  const int mt0 = (p->g[0].M + BM - 1) / BM;
  const int mt1 = p->ngroups > 1 ? (p->g[1].M + BM - 1) / BM : 0;
  const int tiles_m = mt0 + mt1, tiles_n = (p->N + BN - 1) / BN;
  static bool attr_set = false;  // benign race: idempotent
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_ring_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, RLDS_BYTES) != hipSuccess)
      return OMNI_ERR_LAUNCH;
    attr_set = true;
  }
  if (gemm_variant() == 0 && p->K % BK == 0)
    hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(NTHREADS), LDS_BYTES, s, *p, mt0, tiles_m,
                       tiles_n);
  else
    hipLaunchKernelGGL(gemm_bf16_ring_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(NTHREADS), RLDS_BYTES, s, *p, mt0,
                       tiles_m, tiles_n);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

}  // namespace

extern "C" int omni_gemm_bf16(const omni_gemm_params* p, omni_stream stream) {
  if (!p || p->ngroups < 1 || p->ngroups > 2 || p->N <= 0 || p->K <= 0) return OMNI_ERR_BAD_ARG;
  if (p->K % RBK != 0 || p->N % 8 != 0) return OMNI_ERR_UNSUPPORTED;
  for (int g = 0; g < p->ngroups; ++g) {
    const omni_gemm_group& G = p->g[g];
    if (!G.A || !G.W || !G.out || G.M <= 0) return OMNI_ERR_BAD_ARG;
    if (!omni_aligned16(G.A) || !omni_aligned16(G.W) || (G.lda % 8) != 0) return OMNI_ERR_ALIGN;
    if ((reinterpret_cast<uintptr_t>(G.out) & 7) || (G.ldo % 4) != 0) return OMNI_ERR_ALIGN;
    if (p->epilogue == OMNI_EPI_BIAS_GATE_RES) {
      if (!G.res || !G.gate || (!G.row_item_map && G.rows_per_item <= 0)) return OMNI_ERR_BAD_ARG;
      if ((G.ldres % 4) != 0 || (G.gate_item_stride % 4) != 0) return OMNI_ERR_ALIGN;
    }
    if (p->epilogue == OMNI_EPI_BIAS_SPLIT3) {
      if (!G.out1 || !G.out2) return OMNI_ERR_BAD_ARG;
    }
  }
  if (p->epilogue == OMNI_EPI_BIAS_SPLIT3 && (p->split_n <= 0 || p->split_n % 32 != 0 || p->N != 3 * p->split_n))
    return OMNI_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (p->epilogue) {
    case OMNI_EPI_BIAS: return launch<OMNI_EPI_BIAS>(p, s);
    case OMNI_EPI_BIAS_GELU_TANH: return launch<OMNI_EPI_BIAS_GELU_TANH>(p, s);
    case OMNI_EPI_BIAS_GATE_RES: return launch<OMNI_EPI_BIAS_GATE_RES>(p, s);
    case OMNI_EPI_BIAS_SPLIT3: return launch<OMNI_EPI_BIAS_SPLIT3>(p, s);
    default: return OMNI_ERR_BAD_ARG;
  }
}
