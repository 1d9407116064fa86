// VAE-decode kernels for gfx950 (AutoencoderKLQwenImage.decode for one frame).
//
//  * omni_vae_conv2d: 3x3 (pad 1) / 1x1 convolution on v_mfma_f32_32x32x16_bf16.
//    Activations are NHWC bf16, weights [Cout][ky][kx][Cin] (the temporal slice [-1] of the reference's causal
//    Conv3d weights: for a single frame the two zero front-pad frames make the other two slices dead,
//    autoencoder_kl_qwenimage.py:69-84 — executing them, as the reference does, is 3x the MACs).
//    Two kernels behind the one entry point:
//    - conv_bordered_kernel (x_padded && y_padded: the decoder's layers between conv_in and conv_out): activations are
//      zero-bordered rasters and the convolution is a GEMM over row-shifted views of one matrix, LDS-DMA fed — see the
//      comment above that kernel;
//    - conv2d_kernel (everything else: conv_in, conv_out, the encoder, fused x2 upsample / stride-2): an implicit GEMM with
//      the im2col gather in the global->register stage:
//    M = output pixels, N = Cout, K = ks*ks*Cin.  Workgroup tile 128 pixels x (32*NB) channels, 4 waves,
//    each wave 32 pixels x 32*NB channels; swapped MFMA operands so a lane owns one pixel and 4 consecutive
//    channels per register quad (8-byte NHWC stores).  The im2col gather (with optional fused nearest-exact
//    x2 upsample: source = dst >> 1, QwenImageUpsample :112-124) happens in the global->register stage;
//    bias, residual add and clamp are fused in the epilogue.
//  * omni_vae_rmsnorm_silu: y = silu(x / max(||x||_2, 1e-12) * sqrt(C) * gamma) per pixel (QwenImageRMS_norm
//    :108-109 + SiLU), 16 B per lane.
//  * omni_vae_upsample2x_bordered: nearest-exact x2 between zero-bordered rasters.
//  * omni_softmax_rows: in-place row softmax (scale folded) for the single-head mid-block attention.
// Roofline: conv = MFMA-bound for Cin,Cout >= 96; norm/softmax = HBM-bound.
#include "common.h"

namespace {

constexpr int CBM = 128, CBK = 32;

template <int NB>
__global__ __launch_bounds__(256) void conv2d_kernel(const omni_conv_params P) {
  constexpr int CBN = 32 * NB;
  __shared__ __attribute__((aligned(16))) char smem[(CBM + CBN) * CBK * 2];
  char* sA = smem;
  char* sW = smem + CBM * CBK * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int Hout = P.upsample2x ? 2 * P.Hin : (P.downsample2x ? P.Hin / 2 : P.Hin);
  const int Wout = P.upsample2x ? 2 * P.Win : (P.downsample2x ? P.Win / 2 : P.Win);
  const int64_t Mtot = (int64_t)P.B * Hout * Wout;
  const int Ktot = P.ksize * P.ksize * P.Cin;
  const int pad = P.ksize / 2;
  const int64_t m0 = (int64_t)blockIdx.x * CBM;
  const int n0 = blockIdx.y * CBN;

  // staging: A tile 128 rows x 4 chunks(16 B) = 512 chunks -> 2 per thread; W tile CBN x 4 chunks
  int a_b[2], a_oy[2], a_ox[2];
  bool a_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid + 256 * i) >> 2;
    const int64_t m = m0 + row;
    a_ok[i] = m < Mtot;
    const int64_t mm = a_ok[i] ? m : 0;
    a_b[i] = (int)(mm / ((int64_t)Hout * Wout));
    const int rem = (int)(mm - (int64_t)a_b[i] * Hout * Wout);
    a_oy[i] = rem / Wout;
    a_ox[i] = rem - a_oy[i] * Wout;
  }
  const uint16_t* x = P.x;
  const uint16_t* w = P.w;
  u32x4_t areg[2], wreg[(CBN * 4 + 255) / 256];
  auto load_stage = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = (tid + 256 * i) & 3;
      const int k = kt * CBK + c * 8;
      u32x4_t v = {0, 0, 0, 0};
      if (a_ok[i] && k < Ktot) {
        const int tap = k / P.Cin, ci = k - tap * P.Cin;
        const int ky = tap / P.ksize, kx = tap - ky * P.ksize;
        if (P.downsample2x) {
          // QwenImageResample('downsample2d/3d'): ZeroPad2d((0, 1, 0, 1)) then Conv2d(3, stride 2): source = 2*dst + tap,
          // zero beyond the right / bottom edge (autoencoder_kl_qwenimage.py:162-166)
          const int iy = 2 * a_oy[i] + ky, ix = 2 * a_ox[i] + kx;
          if (iy < P.Hin && ix < P.Win)
            v = *reinterpret_cast<const u32x4_t*>(x + (((int64_t)a_b[i] * P.Hin + iy) * P.Win + ix) * P.Cin + ci);
        } else {
        if (P.x_padded) {
          // the input carries a one-pixel zero border ([Hin + 2][Win + 2] raster): no bounds to check
          v = *reinterpret_cast<const u32x4_t*>(x + (((int64_t)a_b[i] * (P.Hin + 2) + a_oy[i] + ky + 1 - pad) * (P.Win + 2) +
                                                     a_ox[i] + kx + 1 - pad) * P.Cin + ci);
        } else {
        const int uy = a_oy[i] + ky - pad, ux = a_ox[i] + kx - pad;
        if (uy >= 0 && uy < Hout && ux >= 0 && ux < Wout) {
          const int iy = P.upsample2x ? (uy >> 1) : uy, ix = P.upsample2x ? (ux >> 1) : ux;
          v = *reinterpret_cast<const u32x4_t*>(x + (((int64_t)a_b[i] * P.Hin + iy) * P.Win + ix) * P.Cin + ci);
        }
        }
        }
      }
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < (CBN * 4 + 255) / 256; ++i) {
      const int id = tid + 256 * i;
      u32x4_t v = {0, 0, 0, 0};
      if (id < CBN * 4) {
        const int n = n0 + (id >> 2), k = kt * CBK + (id & 3) * 8;
        if (n < P.Cout && k < Ktot) v = *reinterpret_cast<const u32x4_t*>(w + (int64_t)n * Ktot + k);
      }
      wreg[i] = v;
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + 256 * i, row = id >> 2, c = id & 3;
      *reinterpret_cast<u32x4_t*>(sA + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < (CBN * 4 + 255) / 256; ++i) {
      const int id = tid + 256 * i, row = id >> 2, c = id & 3;
      if (id < CBN * 4) *reinterpret_cast<u32x4_t*>(sW + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = wreg[i];
    }
  };

  f32x16_t acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

  const int nkt = (Ktot + CBK - 1) / CBK;
  const int arow = wave * 32 + l31;
  load_stage(0);
  for (int kt = 0; kt < nkt; ++kt) {
    store_stage();
    __syncthreads();
    if (kt + 1 < nkt) load_stage(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ks * 2 + hi;
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(sA + arow * 64 + ((ch ^ ((arow >> 2) & 3)) << 4));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int wrow = nb * 32 + l31;
        const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(sW + wrow * 64 + ((ch ^ ((wrow >> 2) & 3)) << 4));
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, acc[nb], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // epilogue: lane owns pixel m, channels n0 + nb*32 + 8q + 4hi + {0..3}
  const int64_t m = m0 + wave * 32 + l31;
  if (m >= Mtot) return;
  const bool do_clamp = P.clamp_lo < P.clamp_hi;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + nb * 32 + q * 8 + hi * 4;
      if (n >= P.Cout) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[nb][q * 4 + j];
      const int nv = min(4, P.Cout - n);
      for (int j = 0; j < nv; ++j) {
        if (P.bias) v[j] += bf16_bits_to_f32(P.bias[n + j]);
        if (P.res) v[j] += bf16_bits_to_f32(P.res[m * P.Cout + n + j]);
        if (do_clamp) v[j] = fminf(fmaxf(v[j], P.clamp_lo), P.clamp_hi);
      }
      uint16_t* dst = P.y + m * P.Cout + n;
      if (nv == 4 && (P.Cout & 3) == 0) {
        u32x2_t o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2_t*>(dst) = o;
      } else {
        for (int j = 0; j < nv; ++j) dst[j] = f32_to_bf16_bits(v[j]);
      }
    }
}

// ---- 3x3 / 1x1 convolution over ZERO-BORDERED rasters: a GEMM with nine shifted A matrices ------------------------------
// Activations carry a one-pixel zero border: [B][Hp = H + 2][Wp = W + 2][C].  Over the bordered raster a 3x3 convolution is
//   y[m] = sum over taps t of  x[m + (ky - 1) * Wp + (kx - 1)] . W_t^T            (m = py * Wp + px, all of the raster)
// i.e. a GEMM whose A operand for k-tile (tap, channel chunk) is the SAME row-major [pixels][Cin] matrix at a constant row
// offset: every A tile is a contiguous block of rows, fetched by LDS-DMA (no gather, no bounds logic: rows before / behind
// the image are outside the buffer descriptor and land as zeros).  Only interior pixels are computed; the kernel also writes
// the zero border of y (the next layer's padding; the norm kernel maps 0 to 0).
// Workgroup = 8 waves (two per SIMD), wave tile 64 pixels x 96 channels (2 x 3 MFMA 32x32x16 tiles, operands swapped so a
// lane owns a pixel and 4 consecutive channels per register quad); WM x WN waves: 4 x 2 (256 px x 192 ch, Cout >= 192) or
// 8 x 1 (512 px x 96 ch).  K-tile = 32 channels of one tap ROW (ky): the A tile holds pixels m0 - 1 .. m0 + MT + 14 of the
// ky-shifted raster and serves the three kx taps at row offsets 0 / 1 / 2 (3 instead of 9 A tiles through L2 per channel
// chunk; 36 MFMAs per wave between barriers).  Two LDS stages, ONE barrier per K-tile (the next K-tile's DMA is issued right
// behind it and lands during this one's 36 MFMAs; a third stage measured no gain).  LDS rows are 64 B with the 16-B chunk index XORed by
// (row >> 2) & 3 (applied on the global-address side of the DMA): ds_read_b128 fragment reads are conflict-free at any
// row offset (16 consecutive rows always hit 16 different (row & 3, chunk) slots).
OMNI_DEVINL u32x4_t conv_srd(const void* base, uint64_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  u32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);     // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xffffffffull ? 0xffffffffull : bytes));
  r[3] = 0x00020000u;
  return r;
}
// LDS-DMA of one 1-KiB piece: lane L -> LDS byte lds_addr + 16 L; the whole (possibly "negative" = wrapped) byte offset is in
// the VGPR, so the descriptor's range check sees all of it
OMNI_DEVINL void conv_dma16(const u32x4_t& srd, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(srd) : "memory");
}

constexpr int CONV_MAX_SEG = 4;                                  // runs per tile
// tuning knobs of the launcher (tools/build_variants.sh builds alternatives for tools/bench_vae.py)
#ifndef OMNI_CONV_BIG_KC
#define OMNI_CONV_BIG_KC 32
#endif
#ifndef OMNI_CONV_BIG_NS
#define OMNI_CONV_BIG_NS 2
#endif
#ifndef OMNI_CONV_SMALL_KC
#define OMNI_CONV_SMALL_KC 32
#endif
#ifndef OMNI_CONV_SMALL_OCC
#define OMNI_CONV_SMALL_OCC 2                                    // workgroups per CU the 256 px x 96 ch tile is compiled for when its LDS allows 3
#endif
// PB = 32-pixel blocks per wave.  PB = 2: K-tile of 32 channels (64-B LDS rows); PB = 4: 16 channels (32-B rows) — the wave
// tile 128 px x 96 ch reads 7 fragments per 12 MFMAs instead of 5 per 6 (the kernel is LDS-read bound: 8 waves x 5 KiB per
// 192 MFMA cycles is 83 % of the LDS pipe) at the same 36 MFMAs per wave between barriers.
template <int WM, int WN, int PB, int KC, int NS>
constexpr int conv_lds_bytes() {
  constexpr int RB = KC * 2, RPP = 1024 / RB;
  return NS * ((32 * PB * WM) / RPP + CONV_MAX_SEG + 3 * ((96 * WN) / RPP)) * 1024;
}
// wait until at most n of this wave's vector-memory operations are outstanding (n wave-uniform, only known at run time: the
// DMA pieces of a K-tile are dealt round-robin over the waves); a smaller immediate than n is always safe
OMNI_DEVINL void conv_wait_vm(int n) {
#define OMNI_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    OMNI_VM(0) OMNI_VM(1) OMNI_VM(2) OMNI_VM(3) OMNI_VM(4) OMNI_VM(5) OMNI_VM(6) OMNI_VM(7) OMNI_VM(8) OMNI_VM(9) OMNI_VM(10)
    OMNI_VM(11) OMNI_VM(12) OMNI_VM(13) OMNI_VM(14) OMNI_VM(15) OMNI_VM(16) OMNI_VM(17) OMNI_VM(18) OMNI_VM(19) OMNI_VM(20)
    OMNI_VM(21) OMNI_VM(22) OMNI_VM(23) OMNI_VM(24) OMNI_VM(25) OMNI_VM(26) OMNI_VM(27) OMNI_VM(28) OMNI_VM(29) OMNI_VM(30)
    OMNI_VM(31) OMNI_VM(32) OMNI_VM(33) OMNI_VM(34) OMNI_VM(35) OMNI_VM(36)
    default: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
  }
#undef OMNI_VM
}

// Tiles are made of RUNS of interior pixels: a run = L consecutive pixels of one image row (L a multiple of the wave's pixel
// count, chosen by the launcher: the row width when it fits), a tile = MT / L runs.  Border pixels are never computed, so
// power-of-two images give whole numbers of tiles per round of 256 CUs (over the full bordered raster 258^2 needs 2.02 rounds
// = 3); the zero border of y is written by the workgroups whose runs touch it.
// KC = channels per K-tile (16: 32-B LDS rows, 32: 64-B rows); NS = LDS stages (K-tiles kt + 1 .. kt + NS - 1 in flight while kt is
// multiplied)
// FUSE: also write P.y_norm = silu?(rmsnorm(y) * P.norm_gamma) — the launcher guarantees ONE channel block (all of a pixel's
// channels in this workgroup: the lane pair of a pixel, times the WN waves that share it)
// UP: x is the HALF-resolution bordered raster [B][Hin / 2 + 2][Win / 2 + 2][Cin] and the conv runs over its nearest-exact x2
// upsample (QwenImageUpsample + the resample conv, autoencoder_kl_qwenimage.py:112-124 / :150-160) without that tensor ever
// existing: P.Hin / P.Win are the OUTPUT's interior size, and the lane that fetches bordered column X of bordered row Y of the
// upsampled raster reads source pixel (((Y - 1) >> 1) + 1, ((X - 1) >> 1) + 1) — border maps to border.
template <int WM, int WN, int PB, int KC, int NS, bool FUSE, bool UP>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN != 4 ? 1 : conv_lds_bytes<WM, WN, PB, KC, NS>() <= 53 * 1024 ? OMNI_CONV_SMALL_OCC
                                                              : conv_lds_bytes<WM, WN, PB, KC, NS>() <= 80 * 1024 ? 2 : 1)) void conv_bordered_kernel(
    const omni_conv_params P, int L) {
  constexpr int NW = WM * WN, MT = 32 * PB * WM, NT = 96 * WN;
  constexpr int RB = KC * 2, RPP = 1024 / RB;                    // LDS row bytes; rows per 1-KiB DMA piece
  constexpr int A_MAX = MT / RPP + CONV_MAX_SEG;                 // every run carries L + RPP rows: pixels x0 - 1 .. x0 + L + RPP - 2
  constexpr int W_PIECES = NT / RPP;                             // per kx tap
  constexpr int STAGE = (A_MAX + 3 * W_PIECES) * 1024;
  static_assert(NS >= 2 && NS * STAGE <= 160 * 1024, "the stages must fit the CU's LDS");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;
  const int Hp = P.Hin + 2, Wp = P.Win + 2;
  const int npix = Hp * Wp;                                      // bordered raster of ONE image
  const int nk = P.ksize, cpt = P.Cin / KC;                      // taps per axis; K-tiles per tap row
  const int nkt = nk * cpt;                                      // K-tile = (ky, channel chunk): all kx taps of it
  const int Ktot = nk * nk * P.Cin;
  const int seg = MT / L, rp = L / RPP + 1;                      // runs per tile; DMA pieces per run
  const int rpr = (P.Win + L - 1) / L, nruns = P.Hin * rpr;      // runs per image row; runs per image
  const int a_pieces = seg * rp;
  // blockIdx.x -> (pixel tile, channel block), XCD-aware: workgroup ids go round-robin over the 8 XCDs (each with its own
  // L2), so XCD k takes the k-th contiguous BAND of the image's tiles — the three ky taps of vertically adjacent runs and the
  // channel blocks of one tile (consecutive ids on one XCD) then meet in ONE L2 instead of being fetched by up to three
  // (measured on 96 -> 96 at 1026^2: 3.3 reads of x from the fabric per element with the plain mapping)
  const int ntiles = (nruns + seg - 1) / seg, nblk = (P.Cout + NT - 1) / NT;
  const int chunk = (ntiles + 7) / 8;                              // tiles per XCD; gridDim.x = 8 * chunk * nblk
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tile = xcd * chunk + idx / nblk;
  if (tile >= ntiles) return;
  const int run0 = tile * seg, n0 = (idx % nblk) * NT, img = blockIdx.z;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  const int Wps = UP ? P.Win / 2 + 2 : Wp;                          // the INPUT raster's row length and pixel count
  const int npix_in = UP ? (P.Hin / 2 + 2) * Wps : npix;
  const u32x4_t x_srd = conv_srd(P.x + (int64_t)img * npix_in * P.Cin, (uint64_t)npix_in * P.Cin * 2);
  const u32x4_t w_srd = conv_srd(P.w, (uint64_t)P.Cout * Ktot * 2);
  // LDS rows hold RB / 16 chunks of 16 B; chunk index XOR swizzle: 64-B rows (r >> 2) & 3, 32-B rows (r >> 3) & 1 — in both
  // cases 16 consecutive rows (at ANY start: the kx taps read at row offsets 0 / 1 / 2) hit 16 different 16-B bank groups.
  // DMA side: lane -> row-in-piece lane / (RB / 16), LDS chunk lane % (RB / 16) holds the logical chunk XORed the same way.
  constexpr int CPR = RB / 16;                                   // chunks per row: 4 or 2
  const int lrow = lane / CPR;
  const uint32_t lc16 = (uint32_t)(((lane % CPR) ^ (KC == 16 ? (lrow >> 3) & 1 : (lrow >> 2) & 3)) * 16);
  const uint32_t a_lane = (uint32_t)(lrow * P.Cin * 2) + lc16;
  const uint32_t w_lane = (uint32_t)(lrow * Ktot * 2) + lc16;
  // raster index of the pixel LEFT of run r's first pixel (runs past the image repeat the last one: computed, never stored)
  auto run_origin = [&](int r) {
    const int run = min(run0 + r, nruns - 1);
    const int y = run / rpr;
    return (y + 1) * Wp + (run - y * rpr) * L;
  };

  // K-tile (ky, cc): per run ONE A block — rows origin + (ky - 1) Wp .. of channel chunk cc; the kx = 0, 1, 2 taps read it at
  // row offsets 0, 1, 2 (a third of the L2 -> LDS traffic of one tile per tap) — and the W rows of its nk taps.
  // The 1-KiB pieces are dealt round-robin over the waves (A pieces g = wave, wave + NW, ..; W pieces likewise); the byte offset
  // of a piece is a per-piece constant (SGPR, computed ONCE: the run / row divisions cost ~100 scalar instructions per piece,
  // 12 per MFMA when redone every K-tile) plus a per-K-tile delta that is the same for every piece of an operand.
  // modulo-2^32 arithmetic: a row before the image wraps to just below 2^32, past the descriptor's range (the launcher keeps
  // the image 16 MiB short of 4 GiB) -> zeros
  constexpr int MAXA = (A_MAX + NW - 1) / NW, MAXW = (3 * W_PIECES + NW - 1) / NW;
  const int w_pieces = nk * W_PIECES;
  const int cntA = (a_pieces - wave + NW - 1) / NW, cntW = (w_pieces - wave + NW - 1) / NW;
  uint32_t preA[MAXA], preW[MAXW];
  uint32_t upx[UP ? MAXA : 1];                                    // UP: per-lane byte offset of the lane's source COLUMN (+ chunk)
  int upy[UP ? MAXA : 1];                                         // UP: bordered output row of the piece's run (centre tap)
#pragma unroll
  for (int i = 0; i < MAXA; ++i) {
    const int g = min(wave + i * NW, a_pieces - 1), r = g / rp, j = g - r * rp;
    if constexpr (UP) {
      const int run = min(run0 + r, nruns - 1), y = run / rpr, X = (run - y * rpr) * L + RPP * j + lrow;
      upx[i] = (uint32_t)((((X - 1) >> 1) + 1) * P.Cin * 2) + lc16;
      upy[i] = __builtin_amdgcn_readfirstlane(y + 1);
      preA[i] = 0u;
    } else {
      preA[i] = __builtin_amdgcn_readfirstlane((uint32_t)(run_origin(r) + RPP * j) * (uint32_t)(P.Cin * 2));
    }
  }
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int q = min(wave + i * NW, w_pieces - 1), kx = q / W_PIECES, rblk = q - kx * W_PIECES;
    preW[i] = __builtin_amdgcn_readfirstlane((uint32_t)(n0 + RPP * rblk) * (uint32_t)(Ktot * 2) + (uint32_t)(kx * P.Cin * 2));
  }
  auto issue = [&](int ky, int cc, int slot) {
    const uint32_t dA = (uint32_t)(nk == 3 ? (ky - 1) * Wp : 0) * (uint32_t)(P.Cin * 2) + (uint32_t)(cc * RB);
    const uint32_t dW = (uint32_t)((ky * nk * P.Cin + cc * KC) * 2);
    const uint32_t sA = lds0 + slot * STAGE + wave * 1024, sW = sA + A_MAX * 1024;
#pragma unroll
    for (int i = 0; i < MAXA; ++i)
      if (i < cntA) {
        if constexpr (UP) {
          const int sy = ((upy[i] + (nk == 3 ? ky - 1 : 0) - 1) >> 1) + 1;
          conv_dma16(x_srd, upx[i] + ((uint32_t)(sy * Wps) * (uint32_t)(P.Cin * 2) + (uint32_t)(cc * RB)), sA + i * (NW * 1024));
        } else {
          conv_dma16(x_srd, a_lane + (preA[i] + dA), sA + i * (NW * 1024));
        }
      }
#pragma unroll
    for (int i = 0; i < MAXW; ++i)
      if (i < cntW) conv_dma16(w_srd, w_lane + (preW[i] + dW), sW + i * (NW * 1024));
  };

  // accumulators start at the bias (lane (l31, hi) holds channels n0 + 96 wn + 32 nb + 8 q + 4 hi + j in acc[.][nb][4 q + j]):
  // the epilogue has no bias traffic
  f32x16_t acc[PB][3];
#pragma unroll
  for (int nb = 0; nb < 3; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wn * 96 + nb * 32 + q * 8 + hi * 4;
      u32x2_t b = {0u, 0u};
      if (P.bias) b = *reinterpret_cast<const u32x2_t*>(P.bias + min(n, P.Cout - 4));
      const float bf[4] = {bf16_lo(b[0]), bf16_hi(b[0]), bf16_lo(b[1]), bf16_hi(b[1])};
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[pb][nb][q * 4 + j] = bf[j];
    }

  auto frag_addr = [&](int r, int c) {
    return (uint32_t)(r * RB + ((c ^ (KC == 16 ? (r >> 3) & 1 : (r >> 2) & 3)) << 4));
  };
  uint32_t w_rd[3][KC / 16];
#pragma unroll
  for (int ks = 0; ks < KC / 16; ++ks)
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) w_rd[nb][ks] = (uint32_t)(A_MAX * 1024) + frag_addr(wn * 96 + nb * 32 + l31, ks * 2 + hi);
  // this wave's 32 PB pixels lie in ONE run (L is a multiple of that): run wr, pixels wi .. of it
  const int wr = (wm * 32 * PB) / L, wi = wm * 32 * PB - wr * L;
  const int a_row0 = wr * (L + RPP) + wi + l31;                  // LDS row of pixel block 0's kx = 0 operand

  int iky = 0, icc = 0;                                          // (ky, chunk) of the next K-tile to issue
  auto issue_next = [&](int kt) {
    issue(iky, icc, kt % NS);
    if (++icc == cpt) { icc = 0; ++iky; }
  };
  const int my_pieces = cntA + cntW;                              // DMA instructions this wave issues per K-tile
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) issue_next(s);
  for (int kt = 0; kt < nkt; ++kt) {
#if defined(OMNI_DEV) && defined(OMNI_CONV_ABL)                  // timing-only ablations (results WRONG): 1 no DMA wait, 2 no DMA after the
    if (!(OMNI_CONV_ABL & 1))                                     // prologue, 4 no MFMA, 8 no barrier
#endif
    conv_wait_vm(min(NS - 2, nkt - 1 - kt) * my_pieces);          // this wave's pieces of K-tile kt have landed (later ones may fly)
#if defined(OMNI_DEV) && defined(OMNI_CONV_ABL)
    if (!(OMNI_CONV_ABL & 8))
#endif
    __syncthreads();                                             // K-tile kt is in LDS; everyone is done with K-tile kt-1's slot
#if defined(OMNI_DEV) && defined(OMNI_CONV_ABL)
    if (!(OMNI_CONV_ABL & 2))
#endif
    if (kt + NS - 1 < nkt) issue_next(kt + NS - 1);               // ... which K-tile kt + NS - 1 now fills
    const char* st = smem + (kt % NS) * STAGE;
    for (int kx = 0; kx < nk; ++kx) {
      const int dx = nk == 3 ? kx : 1;                             // block row 0 is the pixel left of the run
      const char* wst = st + kx * (W_PIECES * 1024);
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        bf16x8_t af[PB], wf[3];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) af[pb] = *reinterpret_cast<const bf16x8_t*>(st + frag_addr(a_row0 + pb * 32 + dx, ks * 2 + hi));
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) wf[nb] = *reinterpret_cast<const bf16x8_t*>(wst + w_rd[nb][ks]);
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) {
#if defined(OMNI_DEV) && defined(OMNI_CONV_ABL)
            if (OMNI_CONV_ABL & 4) { acc[pb][nb][0] += (float)(wf[nb][0] + af[pb][0]); continue; }
#endif
            acc[pb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], af[pb], acc[pb][nb], 0, 0, 0);
          }
      }
    }
  }

  // epilogue: lane (l31, hi) owns one pixel (column of the swapped product) and, per 32-channel block, channels 8 q + 4 hi + {0..3}.
  // v_permlane32_swap between the lane pair of a pixel turns two 8-B quarters into ONE 16-B run (channels 16 h + 8 hi + {0..7}),
  // so a pixel's bias / residual / output move as 16-B accesses; all of a pixel block's residual loads are issued before the
  // first use (addresses clamped instead of branched around: a load inside `if (n < Cout)` is waited for on the spot).
  const bool do_clamp = P.clamp_lo < P.clamp_hi;
  const int my_run = run0 + wr;
  const int crun = min(my_run, nruns - 1);
  const int ry = crun / rpr, rx0 = (crun - ry * rpr) * L;
  const int nch0 = n0 + wn * 96 + hi * 8;                         // the lane's 8-channel runs start at nch0 + 32 nb + 16 h
  u32x4_t kept[FUSE ? PB : 1][3][2];                              // FUSE: the pixel's outputs as rounded to bf16, for the norm pass
  float ss[PB];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int x = rx0 + wi + pb * 32 + l31;
    const bool live = my_run < nruns && x < P.Win;
    const int64_t row = ((int64_t)img * npix + (ry + 1) * Wp + min(x, P.Win - 1) + 1) * P.Cout;
    u32x4_t rs[3][2];
#if defined(OMNI_DEV) && defined(OMNI_CONV_ABL)
    if (OMNI_CONV_ABL & 16) {                                     // 16: no residual loads
#pragma unroll
      for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h) rs[nb][h] = u32x4_t{1u, 2u, 3u, 4u};
    } else
#endif
    if (P.res) {
#pragma unroll
      for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h)                                  // clamped, not branched around (masked at the store)
          rs[nb][h] = *reinterpret_cast<const u32x4_t*>(P.res + row + min(nch0 + nb * 32 + h * 16, P.Cout - 8));
      __builtin_amdgcn_sched_barrier(0);
    }
    ss[pb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                              // quarters q = 2h (first operand) and 2h + 1 (second)
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[pb][nb][(2 * h) * 4 + j]),
                                                           __float_as_uint(acc[pb][nb][(2 * h + 1) * 4 + j]), false, false);
          v[j] = __uint_as_float(sw[0]);
          v[4 + j] = __uint_as_float(sw[1]);
        }
        if (P.res) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[2 * j] += bf16_lo(rs[nb][h][j]);
            v[2 * j + 1] += bf16_hi(rs[nb][h][j]);
          }
        }
        if (do_clamp) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fminf(fmaxf(v[j], P.clamp_lo), P.clamp_hi);
        }
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
        const int nc = nch0 + nb * 32 + h * 16;
        if constexpr (FUSE) {
          if (nc >= P.Cout) o = u32x4_t{0u, 0u, 0u, 0u};
          kept[pb][nb][h] = o;
#pragma unroll
          for (int j = 0; j < 4; ++j) ss[pb] += bf16_lo(o[j]) * bf16_lo(o[j]) + bf16_hi(o[j]) * bf16_hi(o[j]);
          if (P.y && live && nc < P.Cout) *reinterpret_cast<u32x4_t*>(P.y + row + nc) = o;
        } else {
#if defined(OMNI_DEV) && defined(OMNI_CONV_ABL)
          if ((OMNI_CONV_ABL & 32) && o[0] != 0x12345678u) continue;   // 32: no output stores
#endif
          if (live && nc < P.Cout) *reinterpret_cast<u32x4_t*>(P.y + row + nc) = o;
        }
      }
  }
  if constexpr (FUSE) {
    // sum of squares over the pixel's channels: the partner lane (hi ^ 1), then the other waves of the pixel through LDS
    u32x4_t gm[3][2];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
      for (int h = 0; h < 2; ++h) gm[nb][h] = *reinterpret_cast<const u32x4_t*>(P.norm_gamma + min(nch0 + nb * 32 + h * 16, P.Cout - 8));
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ss[pb]), __float_as_uint(ss[pb]), false, false);
      ss[pb] += __uint_as_float(hi ? sw[0] : sw[1]);
    }
    if constexpr (WN > 1) {
      static_assert(WN == 2, "pixel shared by two waves");
      float* red = reinterpret_cast<float*>(smem);               // [WN][MT]
      __syncthreads();                                             // every wave is done with the last K-tile's LDS
      if (hi == 0) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) red[wn * MT + wm * 32 * PB + pb * 32 + l31] = ss[pb];
      }
      __syncthreads();
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) ss[pb] += red[(wn ^ 1) * MT + wm * 32 * PB + pb * 32 + l31];
    }
    const float sqc = sqrtf((float)P.Cout);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int x = rx0 + wi + pb * 32 + l31;
      const bool live = my_run < nruns && x < P.Win;
      const int64_t row = ((int64_t)img * npix + (ry + 1) * Wp + min(x, P.Win - 1) + 1) * P.Cout;
      const float r = sqc / fmaxf(sqrtf(ss[pb]), 1e-12f);
#pragma unroll
      for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x4_t o = kept[pb][nb][h], g = gm[nb][h];
          float f[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f[2 * j] = bf16_lo(o[j]) * r * bf16_lo(g[j]);
            f[2 * j + 1] = bf16_hi(o[j]) * r * bf16_hi(g[j]);
          }
          if (P.norm_silu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
          }
          u32x4_t w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = pack_bf16x2(f[2 * j], f[2 * j + 1]);
          const int nc = nch0 + nb * 32 + h * 16;
          if (live && nc < P.Cout) *reinterpret_cast<u32x4_t*>(P.y_norm + row + nc) = w;
        }
    }
  }

  // the zero border of y, channels [n0, n0 + NT): every run writes the border pixels it touches — the pixel left of a row's
  // first run and right of its last, the stretch above a run of the first image row / below one of the last (with the corners)
  const int c8 = (min(NT, P.Cout - n0)) / 8;
  auto zero_px = [&](int m_first, int count) {
    for (int i = tid; i < count * c8; i += 64 * NW) {
      const int px = i / c8, c = i - px * c8;
      const int64_t at = ((int64_t)img * npix + m_first + px) * P.Cout + n0 + c * 8;
      if (!FUSE || P.y) *reinterpret_cast<u32x4_t*>(P.y + at) = u32x4_t{0u, 0u, 0u, 0u};
      if constexpr (FUSE) *reinterpret_cast<u32x4_t*>(P.y_norm + at) = u32x4_t{0u, 0u, 0u, 0u};
    }
  };
  for (int r = 0; r < seg; ++r) {
    const int run = run0 + r;
    if (run >= nruns) break;
    const int y = run / rpr, x0 = (run - y * rpr) * L, len = min(L, P.Win - x0);
    const bool first = x0 == 0, last = x0 + len == P.Win;
    if (first) zero_px((y + 1) * Wp, 1);
    if (last) zero_px((y + 1) * Wp + P.Win + 1, 1);
    if (y == 0) zero_px(x0 + 1 - (first ? 1 : 0), len + (first ? 1 : 0) + (last ? 1 : 0));
    if (y == P.Hin - 1) zero_px((Hp - 1) * Wp + x0 + 1 - (first ? 1 : 0), len + (first ? 1 : 0) + (last ? 1 : 0));
  }
}

// ---- 3x3 convolution from a zero-bordered raster to a FEW output channels (the decoder's conv_out: 96 -> 3 at full image
// resolution, autoencoder_kl_qwenimage.py:737-739) -------------------------------------------------------------------------
// 2 * 864 * 3 flop per output pixel against 192 B of input: HBM-bound (202 MB in at 1024^2).  The general kernels pad the
// three channels to a 32-wide tile and gather A per K-tile through LDS (0.23 ms at 1024^2); here a wave owns 64 consecutive
// pixels of an image row (four 16-pixel MFMA 16x16x32 row blocks, the <= 16 output channels are the MFMA's 16 columns) and
// takes its A fragments straight from global memory — a lane's 8 consecutive channels of pixel (y + ky, x + kx) are 16
// contiguous bytes of the bordered raster, no bounds logic — one tap ahead of the MFMAs; the weights (16 x 9 Cin bf16, zero
// rows behind Cout) sit in LDS in fragment order.  Output: plain raster [B][H][W][Cout].
constexpr int CF_MAX_CIN = 128;
template <int NCC>                                                // Cin / 32
__global__ __launch_bounds__(256) void conv_few_kernel(const omni_conv_params P) {
  __shared__ __attribute__((aligned(16))) uint16_t wl[9 * NCC * 64 * 8];   // [tap][chunk][lane][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  constexpr int ncc = NCC;
  const int Wp = P.Win + 2;
  const int segs = (P.Win + 255) / 256;
  const int seg = blockIdx.x % segs, yrow = blockIdx.x / segs, img = blockIdx.y;
  // weights -> LDS in B-fragment order: lane (n = l15, g) of step (tap, cc) holds w[n][tap][cc * 32 + 8 g .. + 8]
  for (int i = tid; i < 9 * ncc * 64; i += 256) {
    const int ln = i & 63, st = i >> 6, tap = st / ncc, cc = st - tap * ncc, n = ln & 15, gg = ln >> 4;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (n < P.Cout) v = *reinterpret_cast<const u32x4_t*>(P.w + ((int64_t)n * 9 + tap) * P.Cin + cc * 32 + gg * 8);
    *reinterpret_cast<u32x4_t*>(wl + (int64_t)i * 8) = v;
  }
  __syncthreads();
  const int x0 = seg * 256 + wave * 64;
  if (x0 >= P.Win) return;
  const float bv = (P.bias && l15 < P.Cout) ? bf16_bits_to_f32(P.bias[l15]) : 0.f;
  f32x4_t acc[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) acc[mb] = f32x4_t{bv, bv, bv, bv};
  // lane's pixel of row block mb: x0 + 16 mb + l15 (clamped into the row: clamped pixels are computed, never stored)
  const uint16_t* xb = P.x + ((int64_t)img * (P.Hin + 2) + yrow) * Wp * P.Cin + g * 8;
  int px[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) px[mb] = min(x0 + mb * 16 + l15, P.Win - 1);
  constexpr int MAXCC = NCC;
  bf16x8_t a[2][MAXCC][4];
  auto load_tap = [&](int tap, bf16x8_t (&dst)[MAXCC][4]) {
    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
    for (int cc = 0; cc < MAXCC; ++cc)
      if (cc < ncc) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
          dst[cc][mb] = *reinterpret_cast<const bf16x8_t*>(xb + ((int64_t)ky * Wp + px[mb] + kx) * P.Cin + cc * 32);
      }
  };
  load_tap(0, a[0]);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    if (tap + 1 < 9) load_tap(tap + 1, a[(tap + 1) & 1]);
#pragma unroll
    for (int cc = 0; cc < MAXCC; ++cc)
      if (cc < ncc) {
        const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(wl + ((int64_t)(tap * ncc + cc) * 64 + lane) * 8);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tap & 1][cc][mb], wf, acc[mb], 0, 0, 0);
      }
  }
  // D[row = 4 g + j (pixel of the block)][col = l15 (channel)]
  if (l15 < P.Cout) {
    const bool do_clamp = P.clamp_lo < P.clamp_hi;
    uint16_t* yb = P.y + (((int64_t)img * P.Hin + yrow) * P.Win) * P.Cout + l15;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = x0 + mb * 16 + g * 4 + j;
        float v = acc[mb][j];
        if (do_clamp) v = fminf(fmaxf(v, P.clamp_lo), P.clamp_hi);
        if (x < P.Win) yb[(int64_t)x * P.Cout] = f32_to_bf16_bits(v);
      }
  }
}

// nearest-exact x2 upsample between zero-bordered rasters (QwenImageUpsample, autoencoder_kl_qwenimage.py:112-124):
// y[oy + 1][ox + 1] = x[(oy >> 1) + 1][(ox >> 1) + 1], border zero.  16 B per lane, HBM-bound.
__global__ __launch_bounds__(256) void upsample2x_bordered_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                                  int B, int H, int W, int C) {
  const int c8 = C / 8, Ho = 2 * H + 2, Wo = 2 * W + 2;
  const int64_t total = (int64_t)B * Ho * Wo * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c8);
    const int64_t p = i / c8;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), b = (int)(p / ((int64_t)Wo * Ho));
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (oy >= 1 && oy <= 2 * H && ox >= 1 && ox <= 2 * W)
      v = *reinterpret_cast<const u32x4_t*>(x + ((((int64_t)b * (H + 2) + ((oy - 1) >> 1) + 1) * (W + 2)) + ((ox - 1) >> 1) + 1) * C + c * 8);
    *reinterpret_cast<u32x4_t*>(y + p * C + c * 8) = v;
  }
}

// channel RMS-norm (+SiLU) over NHWC pixels: C/8 lanes per pixel (C % 8 == 0, C <= 512 -> <= 64 lanes)
template <int LPP>  // lanes per pixel (power of two >= C/8)
__global__ __launch_bounds__(256) void vae_rmsnorm_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                          int64_t rows, int Cc, const uint16_t* __restrict__ gamma,
                                                          int silu) {
  const int sub = threadIdx.x % LPP;
  const int64_t row = (int64_t)blockIdx.x * (256 / LPP) + threadIdx.x / LPP;
  if (row >= rows) return;
  const bool act = sub * 8 < Cc;
  float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (act) {
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(x + row * Cc + sub * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = bf16_lo(w[i]);
      f[2 * i + 1] = bf16_hi(w[i]);
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
  ss = wave_sum<LPP>(ss);
  const float r = sqrtf((float)Cc) / fmaxf(sqrtf(ss), 1e-12f);
  if (act) {
    const u32x4_t g = *reinterpret_cast<const u32x4_t*>(gamma + sub * 8);
    float o[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = f[2 * i] * r * bf16_lo(g[i]);
      o[2 * i + 1] = f[2 * i + 1] * r * bf16_hi(g[i]);
    }
    if (silu) {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = silu_f(o[i]);
    }
    u32x4_t w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
    *reinterpret_cast<u32x4_t*>(y + row * Cc + sub * 8) = w;
  }
}

// ---- single-head attention of the VAE mid block (head dim = channels = 384) as ONE flash kernel ----------------------------
// QwenImageAttentionBlock.forward (autoencoder_kl_qwenimage.py:305-330): F.scaled_dot_product_attention over all H*W tokens of an
// image with one head of 384 channels.  As GEMM -> softmax -> GEMM the 16384 x 16384 (1024^2) score matrix crosses HBM four
// times (1 GiB per image; 17 GB at 2048^2); here it never leaves the registers.
// Workgroup = 4 waves = 128 queries of one image, ONE wave per SIMD with the whole register file: a wave owns 32 queries, its
// O^T accumulators are 12 MFMA 32x32 blocks (384 channels x 32 queries = 192 registers), Q~ (24 fragments = 96 registers) stays
// in registers for the whole kernel.  KV tile = 32 keys (K 24 KiB + V 24 KiB per LDS stage, two stages): per tile 24 MFMA
// 32x32x16 for S^T = K Q^T (two accumulator chains) and 24 for O^T += V^T P^T.  Operands swapped so that a lane owns ONE query:
// online softmax is lane-local (one v_permlane32_swap for the cross-half max), P^T stays in registers as the B operand, V^T
// fragments come from hardware-transposed reads (ds_read_b64_tr_b16), K fragments from 64-B column blocks with XOR-swizzled
// 16-B chunks (conflict-free ds_read_b128).  Tiles are staged by LDS-DMA one tile ahead
// (landing under the tile's MFMAs); fragment reads are hand-issued with counted waits (hipcc retires every read
// with lgkmcnt(0)).  Same structure as round 1's flash_attn_fwd_kernel (attention.hip), widened to 384.
constexpr int VA_DH = 384, VA_KV = 32, VA_NW = 4;
constexpr int VA_K_BYTES = VA_KV * VA_DH * 2;                      // 24 KiB
constexpr int VA_STAGE = 2 * VA_K_BYTES;                           // K + V
constexpr int VA_LDS = 2 * VA_STAGE;                               // 96 KiB

template <int OFF>
OMNI_DEVINL bf16x8_t va_read16(uint32_t addr) {                    // invisible to hipcc's waitcnt pass: waits are counted by hand
  bf16x8_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
template <int OFF>
OMNI_DEVINL u32x2_t va_tr_read8(uint32_t addr) {                    // ds_read_b64_tr_b16: hardware 4x4 transpose read
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}

__global__ __launch_bounds__(VA_NW * 64, 1) void vae_attn_fwd_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                                    const uint16_t* __restrict__ v, uint16_t* __restrict__ out,
                                                                    int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int seq_len,
                                                                    float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  constexpr int QBLK = 32 * VA_NW;
  const int qb = blockIdx.x, img = blockIdx.y;
  q += (int64_t)img * seq_len * ldq;
  out += (int64_t)img * seq_len * ldo;
  const uint16_t* kbase = k + (int64_t)img * seq_len * ldk;
  const uint16_t* vbase = v + (int64_t)img * seq_len * ldv;

  // ---- Q fragments (B operand): lane holds q = l31, d = ks*16 + hi*8 .. +8 ----------------------------------------------------
  bf16x8_t qf[24];
  {
    const int qrow = min(qb * QBLK + wave * 32 + l31, seq_len - 1);
    const uint16_t* qp = q + (int64_t)qrow * ldq + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
    // Q~ lives in AGPRs: the MFMAs below are asm statements whose B operand has the "a" constraint.  (hipcc's own MFMAs take A / B
    // from arch VGPRs only: it kept all of Q~ in the 256 VGPRs, spilled half of it, and the scratch reloads' vmcnt waits drained
    // the LDS-DMA in flight every tile: 10x slower.)
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) asm volatile("" : "+a"(qf[ks]));
  }
  // register file of a wave (512): AGPRs = Q~ (96) + O^T blocks 0..9 (160); VGPRs = O^T blocks 10, 11 (32), S (48), fragments, addresses

  // ---- staging by LDS-DMA (buffer_load ... lds: lane L of piece p lands at LDS bytes p * 1024 + 16 L of the operand's image).
  // Both images are made of 64-B (32-channel) column blocks: piece p = column block p / 2, keys (p % 2) * 16 + L / 4, 16-B chunk
  // L % 4 of the block — so a lane's source offset is ONE per-lane constant plus a per-piece uniform, for K and for V alike:
  //   K image [12 column blocks][32 keys][64 B], the chunk index XORed with (key >> 2) & 3 (on the source side): a ds_read_b128
  //           fragment read (16 lanes = 16 keys, 64 B apart) then hits 16 different 16-B bank groups;
  //   V image [12 column blocks][8 groups of 4 keys][4 keys][64 B]: the tr-read layout of attention.hip, 32 keys per block.
  // rows past the end of the image are outside the descriptors' range and land as 0 (those keys are masked to -inf)
  const int dkey = lane >> 2, dpos = lane & 3;
  const uint32_t k_ld_lane = (uint32_t)(dkey * ldk * 2 + ((dpos ^ ((dkey >> 2) & 3)) << 4));
  const uint32_t v_ld_lane = (uint32_t)(dkey * ldv * 2 + (dpos << 4));
  const u32x4_t k_srd = conv_srd(kbase, (uint64_t)((int64_t)(seq_len - 1) * ldk * 2 + VA_DH * 2));
  const u32x4_t v_srd = conv_srd(vbase, (uint64_t)((int64_t)(seq_len - 1) * ldv * 2 + VA_DH * 2));
  const uint32_t k_tile_stride = (uint32_t)(VA_KV * ldk * 2), v_tile_stride = (uint32_t)(VA_KV * ldv * 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // piece j of this wave's twelve per tile (j < 6: K piece 6 wave + j, else V piece 6 wave + j - 6) of tile t -> stage t & 1
  auto dma_piece = [&](int t, int j) {
    const int i = j < 6 ? j : j - 6, p = wave * 6 + i;
    const uint32_t dst = lds0 + (t & 1) * VA_STAGE + (wave * 6 + i) * 1024;
    if (j < 6) {
      const uint32_t ku = __builtin_amdgcn_readfirstlane((uint32_t)t * k_tile_stride + (uint32_t)((p & 1) * 16) * (uint32_t)(ldk * 2) + (uint32_t)((p >> 1) * 64));
      conv_dma16(k_srd, k_ld_lane + ku, dst);
    } else {
      const uint32_t vu = __builtin_amdgcn_readfirstlane((uint32_t)t * v_tile_stride + (uint32_t)((p & 1) * 16) * (uint32_t)(ldv * 2) + (uint32_t)((p >> 1) * 64));
      conv_dma16(v_srd, v_ld_lane + vu, dst + VA_K_BYTES);
    }
  };
  auto dma_tile = [&](int t) {
#pragma unroll
    for (int j = 0; j < 12; ++j) dma_piece(t, j);
  };

  // ---- fragment read addresses ---------------------------------------------------------------------------------------------
  // K, k-step ks (chunk ks * 2 + hi of row l31): column block ks >> 1 (an immediate: 2048 B each), chunk (ks & 1) * 2 + hi swizzled
  uint32_t k_addr[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) k_addr[e] = l31 * 64 + ((((uint32_t)(e * 2 + hi)) ^ ((l31 >> 2) & 3)) << 4);
  // V (tr read): m = lane & 15, g = lane >> 4:  (m >> 2) * 64 + (g & 1) * 32 + (m & 3) * 8 + hi * 256
  const uint32_t v_lane_off = VA_K_BYTES + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8 + hi * 256;

  f32x16_t o[12];
#pragma unroll
  for (int d = 0; d < 12; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;

  const int ntiles = (seq_len + VA_KV - 1) / VA_KV;
  dma_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const int kv0 = t * VA_KV;
    const bool more = t + 1 < ntiles;                              // tile t + 1's DMA pieces go out one per two MFMAs of the S phase (a
                                                                   // piece costs its wave ~60 issue cycles: twelve in a burst idle the pipe)

    // ---- S^T = K Q^T: 24 MFMAs on two accumulator chains; K fragments are read 4 deep ahead of their MFMA
    f32x16_t s0, s1;                                               // (the chains open with C = 0 as an inline constant: no VALU
                                                                   //  write of an accumulator right in front of an asm MFMA)
    {
      const uint32_t kst = lds0 + cur * VA_STAGE;
      bf16x8_t kf[4];
#define OMNI_VA_KREAD(i) kf[(i) & 3] = va_read16<0>(k_addr[(i) & 1] + kst + (uint32_t)(((i) >> 1) * 2048))
      OMNI_VA_KREAD(0); OMNI_VA_KREAD(1); OMNI_VA_KREAD(2); OMNI_VA_KREAD(3);
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        if (i <= 20) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else if (i == 21) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if (i == 22) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (i == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s0) : "v"(kf[i & 3]), "a"(qf[i]));
        else if (i == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s1) : "v"(kf[i & 3]), "a"(qf[i]));
        else if (i & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s1) : "v"(kf[i & 3]), "a"(qf[i]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0) : "v"(kf[i & 3]), "a"(qf[i]));
        __builtin_amdgcn_sched_barrier(0);
        if (i + 4 < 24) OMNI_VA_KREAD(i + 4);
        if ((i & 1) && more) dma_piece(t + 1, i >> 1);               // (its stage was drained one barrier ago)
      }
#undef OMNI_VA_KREAD
    }
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s0), "+v"(s1));     // the last MFMAs' passes before a VALU reader (no interlock)
    f32x16_t s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = s0[i] + s1[i];
    // ---- mask the ragged tail (last tile only; wave-uniform branch): row r of the block = key (r & 3) + 8 (r >> 2) + 4 hi
    if (kv0 + VA_KV > seq_len) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kv0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= seq_len) s[r] = -INFINITY;
    }
    // ---- online softmax, lane-local (one cross-half exchange for the max); defer-max as in attention.hip
    bf16x8_t pf[2];
    {
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      constexpr float DEFER = 6.0f;
      if (!__all((mx - m_run) * scale_log2e <= DEFER)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = m_run == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 12; ++d) {                             // one block at a time: all 192 values at once in VGPR temporaries
#pragma unroll                                                     // push the loop's long-lived values (the DMA offsets) to scratch
          for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      const float mneg = -m_run * scale_log2e;
      float psum = 0.0f;
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[ss * 8 + e], scale_log2e, mneg));
          psum += p;
          pf[ss][e] = (__bf16)p;
        }
      l_run += psum;
    }
    // ---- O^T += V^T P^T: 24 MFMAs, i -> (ss = i / 12, d = i % 12); a V^T fragment = two transposed reads, fetched three ahead
    {
      const uint32_t vb = lds0 + cur * VA_STAGE + v_lane_off;
      u32x2_t vlo[4], vhi[4];
#define OMNI_VA_VOFF(i) (((i) % 12) * (VA_KV / 4 * 256) + ((i) / 12) * 1024)
#define OMNI_VA_VREAD(i)                                        \
  do {                                                          \
    vlo[(i) & 3] = va_tr_read8<OMNI_VA_VOFF(i)>(vb);            \
    vhi[(i) & 3] = va_tr_read8<OMNI_VA_VOFF(i) + 512>(vb);      \
  } while (0)
#define OMNI_VA_PV(i)                                                                                            \
  do {                                                                                                           \
    if ((i) <= 21) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                            \
    else if ((i) == 22) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                                       \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    {                                                                                                            \
      const u32x4_t w_ = {vlo[(i) & 3][0], vlo[(i) & 3][1], vhi[(i) & 3][0], vhi[(i) & 3][1]};                   \
      if ((i) % 12 < 10) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[(i) % 12]) : "v"(w_), "v"(pf[(i) / 12])); \
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(o[(i) % 12]) : "v"(w_), "v"(pf[(i) / 12]));       \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  } while (0)
      OMNI_VA_VREAD(0); OMNI_VA_VREAD(1); OMNI_VA_VREAD(2);
      asm volatile("s_nop 3" : "+v"(pf[0]), "+v"(pf[1]));           // P^T was written by VALU converts: wait states before an asm MFMA reads it
      OMNI_VA_PV(0);  OMNI_VA_VREAD(3);  OMNI_VA_PV(1);  OMNI_VA_VREAD(4);  OMNI_VA_PV(2);  OMNI_VA_VREAD(5);
      OMNI_VA_PV(3);  OMNI_VA_VREAD(6);  OMNI_VA_PV(4);  OMNI_VA_VREAD(7);  OMNI_VA_PV(5);  OMNI_VA_VREAD(8);
      OMNI_VA_PV(6);  OMNI_VA_VREAD(9);  OMNI_VA_PV(7);  OMNI_VA_VREAD(10); OMNI_VA_PV(8);  OMNI_VA_VREAD(11);
      OMNI_VA_PV(9);  OMNI_VA_VREAD(12); OMNI_VA_PV(10); OMNI_VA_VREAD(13); OMNI_VA_PV(11); OMNI_VA_VREAD(14);
      OMNI_VA_PV(12); OMNI_VA_VREAD(15); OMNI_VA_PV(13); OMNI_VA_VREAD(16); OMNI_VA_PV(14); OMNI_VA_VREAD(17);
      OMNI_VA_PV(15); OMNI_VA_VREAD(18); OMNI_VA_PV(16); OMNI_VA_VREAD(19); OMNI_VA_PV(17); OMNI_VA_VREAD(20);
      OMNI_VA_PV(18); OMNI_VA_VREAD(21); OMNI_VA_PV(19); OMNI_VA_VREAD(22); OMNI_VA_PV(20); OMNI_VA_VREAD(23);
      OMNI_VA_PV(21); OMNI_VA_PV(22); OMNI_VA_PV(23);
#undef OMNI_VA_PV
#undef OMNI_VA_VREAD
#undef OMNI_VA_VOFF
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's pieces of tile t + 1
    __syncthreads();
  }

  // ---- epilogue: O[q][d] = O^T / l; lane holds q = l31, d = dblk*32 + 8*qd + 4*hi + {0..3}
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");                 // the last P.V MFMAs' passes before the VALU reads O^T
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = qb * QBLK + wave * 32 + l31;
  if (qrow < seq_len) {
    uint16_t* op = out + (int64_t)qrow * ldo + hi * 4;
#pragma unroll
    for (int d = 0; d < 12; ++d)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        u32x2_t w;
        w[0] = pack_bf16x2(o[d][qd * 4 + 0] * inv, o[d][qd * 4 + 1] * inv);
        w[1] = pack_bf16x2(o[d][qd * 4 + 2] * inv, o[d][qd * 4 + 3] * inv);
        *reinterpret_cast<u32x2_t*>(op + d * 32 + qd * 8) = w;
      }
  }
}

// in-place softmax over rows of `cols` bf16 scores: p = exp((s - max) * scale) / sum.  One workgroup per row.
// softmax_rows_reg_kernel: the row stays in registers between the passes (cols <= 64 * THREADS: 8 chunks of 8 per thread, all
// loads issued before the first use) — one read and one write of the scores (HBM-bound: the VAE's 16384 x 16384 mid-block
// attention moves 1 GiB through here); softmax_rows_kernel (any length) re-reads the row for each of its three passes.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void softmax_rows_reg_kernel(uint16_t* __restrict__ s, int64_t ld, int cols, float scale) {
  constexpr int NW = THREADS / 64, CH = 8;
  __shared__ float red[2 * NW];
  uint16_t* row = s + (int64_t)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch = cols / 8;
  u32x4_t w[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) w[i] = *reinterpret_cast<const u32x4_t*>(row + min(tid + i * THREADS, nch - 1) * 8);
  __builtin_amdgcn_sched_barrier(0);
  float e[CH][8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const bool in = tid + i * THREADS < nch;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      e[i][2 * j] = in ? bf16_lo(w[i][j]) : -INFINITY;
      e[i][2 * j + 1] = in ? bf16_hi(w[i][j]) : -INFINITY;
      mx = fmaxf(mx, fmaxf(e[i][2 * j], e[i][2 * j + 1]));
    }
  }
  mx = wave_max<64>(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) mx = fmaxf(mx, red[k]);
  const float c2 = scale * 1.4426950408889634f;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      e[i][j] = __builtin_amdgcn_exp2f((e[i][j] - mx) * c2);     // (-inf - mx) * c2 = -inf -> 0
      sum += e[i][j];
    }
  sum = wave_sum<64>(sum);
  if (lane == 0) red[NW + wave] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < NW; ++k) tot += red[NW + k];
  const float inv = 1.0f / tot;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    if (tid + i * THREADS >= nch) continue;
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(e[i][2 * j] * inv, e[i][2 * j + 1] * inv);
    *reinterpret_cast<u32x4_t*>(row + (tid + i * THREADS) * 8) = o;
  }
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(uint16_t* __restrict__ s, int64_t ld, int cols,
                                                           float scale) {
  __shared__ float red[8];
  uint16_t* row = s + (int64_t)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch = cols / 8;
  float mx = -INFINITY;
  for (int c = tid; c < nch; c += 256) {
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(row + c * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) mx = fmaxf(mx, fmaxf(bf16_lo(w[i]), bf16_hi(w[i])));
  }
  mx = wave_max<64>(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float c2 = scale * 1.4426950408889634f;
  float sum = 0.f;
  for (int c = tid; c < nch; c += 256) {
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(row + c * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      sum += __builtin_amdgcn_exp2f((bf16_lo(w[i]) - mx) * c2) + __builtin_amdgcn_exp2f((bf16_hi(w[i]) - mx) * c2);
  }
  sum = wave_sum<64>(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = tid; c < nch; c += 256) {
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(row + c * 8);
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      o[i] = pack_bf16x2(__builtin_amdgcn_exp2f((bf16_lo(w[i]) - mx) * c2) * inv,
                         __builtin_amdgcn_exp2f((bf16_hi(w[i]) - mx) * c2) * inv);
    *reinterpret_cast<u32x4_t*>(row + c * 8) = o;
  }
}

}  // namespace

namespace {
// bordered rasters: tile geometry of the launch (run length: a multiple of the wave's pixel count `gran`, the row width when it
// fits the tile, tile / L <= CONV_MAX_SEG)
int conv_run_len(const omni_conv_params* p, int mt, int gran) {
  int L = (p->Win + gran - 1) / gran * gran;
  if (L > mt) L = mt;
  while (mt % L || mt / L > CONV_MAX_SEG) L += gran;              // (mt / gran is a power of two: terminates at mt)
  return L;
}
int64_t conv_tiles_of(const omni_conv_params* p, int mt, int gran) {
  const int L = conv_run_len(p, mt, gran);
  const int64_t runs = (int64_t)p->Hin * ((p->Win + L - 1) / L);
  return (runs + mt / L - 1) / (mt / L);
}
// Cout >= 192: 128-pixel waves (8 waves = 512 px x 192 ch, one workgroup per CU) when they fill the chip at least once
// (+10 % over the small tiles at 258^2 / 514^2); else four-wave workgroups of 256 px x 96 ch, two per CU (for Cout = 96 the
// 8 x 1 arrangement of 128-pixel waves measured 443 vs 568 TF/s at 1026^2: the K loop is only nine K-tiles long there)
bool conv_uses_big_tile(const omni_conv_params* p) {
  return p->Cout >= 192 && conv_tiles_of(p, 512, 128) * ((p->Cout + 191) / 192) * p->B >= 256;
}
bool conv_fuses_norm(const omni_conv_params* p) {
  if (!p->norm_gamma || !p->y_padded || !p->x_padded) return false;
  return conv_uses_big_tile(p) ? p->Cout <= 192 : p->Cout <= 96;
}
}  // namespace

namespace {
// bordered rasters with upsample2x: the kernel (and the tile geometry) see the OUTPUT's interior size in Hin / Win
omni_conv_params conv_output_sized(const omni_conv_params* p) {
  omni_conv_params q = *p;
  if (p->upsample2x && p->x_padded && p->y_padded) {
    q.Hin = 2 * p->Hin;
    q.Win = 2 * p->Win;
  }
  return q;
}
}  // namespace

extern "C" int omni_vae_conv2d_fuses_norm(const omni_conv_params* p) {
  if (!p) return 0;
  const omni_conv_params q = conv_output_sized(p);
  return conv_fuses_norm(&q) ? 1 : 0;
}

extern "C" int omni_vae_conv2d(const omni_conv_params* p, omni_stream stream) {
  if (!p || !p->x || !p->w || p->B <= 0 || p->Hin <= 0 || p->Win <= 0 || p->Cin <= 0 || p->Cout <= 0) return OMNI_ERR_BAD_ARG;
  if (p->norm_gamma && !p->y_norm) return OMNI_ERR_BAD_ARG;
  const omni_conv_params q = conv_output_sized(p);
  const bool fuse = conv_fuses_norm(&q);
  if (!p->y && !fuse) return OMNI_ERR_BAD_ARG;
  if ((p->ksize != 1 && p->ksize != 3) || p->Cin % 8) return OMNI_ERR_UNSUPPORTED;
  if (p->gamma) return OMNI_ERR_UNSUPPORTED;  // fused norm prologue: not built yet (use omni_vae_rmsnorm_silu)
  if (p->downsample2x && (p->upsample2x || p->ksize != 3 || (p->Hin & 1) || (p->Win & 1))) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(p->x) || !omni_aligned16(p->w)) return OMNI_ERR_ALIGN;
  const int Hout = p->upsample2x ? 2 * p->Hin : (p->downsample2x ? p->Hin / 2 : p->Hin);
  const int Wout = p->upsample2x ? 2 * p->Win : (p->downsample2x ? p->Win / 2 : p->Win);
  const int64_t M = (int64_t)p->B * Hout * Wout;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // norm_gamma on a shape the conv kernel cannot norm itself: the separate pass, from y (0 -> 0: a zero border stays zero)
  auto norm_after = [&]() -> int {
    if (!p->norm_gamma || fuse) return OMNI_OK;
    const int64_t rows = (int64_t)p->B * (Hout + 2 * (p->y_padded ? 1 : 0)) * (Wout + 2 * (p->y_padded ? 1 : 0));
    return omni_vae_rmsnorm_silu(p->y, p->y_norm, rows, p->Cout, p->norm_gamma, p->norm_silu, stream);
  };
  if (p->y_padded) {
    // zero-bordered rasters in and out: the shifted-GEMM kernel (upsample2x: over the x2 upsample of x, 3x3 only, no residual)
    if (!p->x_padded || p->downsample2x || p->Cin % 32 || p->Cout % 8) return OMNI_ERR_UNSUPPORTED;
    if (p->upsample2x && (p->ksize != 3 || p->res)) return OMNI_ERR_UNSUPPORTED;
    if (!omni_aligned16(p->y) || (p->bias && (reinterpret_cast<uintptr_t>(p->bias) & 7)) || (p->res && !omni_aligned16(p->res)) ||
        (fuse && (!omni_aligned16(p->y_norm) || !omni_aligned16(p->norm_gamma))))
      return OMNI_ERR_ALIGN;
    const int64_t npix = (int64_t)(p->Hin + 2) * (p->Win + 2);       // the INPUT raster
    if (npix * p->Cin * 2 >= (1ll << 32) - (1 << 24) || p->B > 65535) return OMNI_ERR_UNSUPPORTED;   // 32-bit DMA offsets per image
    auto launch = [&]<int WM, int WN, int PB, int KC, int NS, bool FUSE, bool UP>() -> int {
      constexpr int lds = conv_lds_bytes<WM, WN, PB, KC, NS>(), MT = 32 * PB * WM, NT = 96 * WN;
      static std::atomic<uint64_t> attr_done{0};     // per device and per template instance (common.h omni_once_per_device)
      OMNI_TRY_STATUS(omni_once_per_device(attr_done, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bordered_kernel<WM, WN, PB, KC, NS, FUSE, UP>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
      }));
      const int64_t chunk = (conv_tiles_of(&q, MT, 32 * PB) + 7) / 8;   // tiles per XCD (see the kernel's blockIdx mapping)
      hipLaunchKernelGGL((conv_bordered_kernel<WM, WN, PB, KC, NS, FUSE, UP>), dim3((unsigned)(8 * chunk * ((q.Cout + NT - 1) / NT)), 1, q.B),
                         dim3(64 * WM * WN), lds, s, q, conv_run_len(&q, MT, 32 * PB));
      return OMNI_OK;
    };
    auto pick = [&]<bool FUSE, bool UP>() -> int {
      return conv_uses_big_tile(&q) ? launch.template operator()<4, 2, 4, OMNI_CONV_BIG_KC, OMNI_CONV_BIG_NS, FUSE, UP>()
                                    : launch.template operator()<4, 1, 2, OMNI_CONV_SMALL_KC, 2, FUSE, UP>();
    };
    const int rc = p->upsample2x ? (fuse ? pick.template operator()<true, true>() : pick.template operator()<false, true>())
                                 : (fuse ? pick.template operator()<true, false>() : pick.template operator()<false, false>());
    if (rc != OMNI_OK) return rc;
    OMNI_CHECK_LAUNCH();
    return norm_after();
  }
  if (p->x_padded && (p->upsample2x || p->downsample2x)) return OMNI_ERR_UNSUPPORTED;
  if (p->x_padded && p->ksize == 3 && p->Cout <= 16 && p->Cin % 32 == 0 && p->Cin <= CF_MAX_CIN && !p->res && p->B <= 65535 &&
      (int64_t)p->Hin * ((p->Win + 255) / 256) < (1ll << 31)) {
    const dim3 grid((unsigned)(p->Hin * ((p->Win + 255) / 256)), p->B);
    switch (p->Cin / 32) {
      case 1: hipLaunchKernelGGL(conv_few_kernel<1>, grid, dim3(256), 0, s, *p); break;
      case 2: hipLaunchKernelGGL(conv_few_kernel<2>, grid, dim3(256), 0, s, *p); break;
      case 3: hipLaunchKernelGGL(conv_few_kernel<3>, grid, dim3(256), 0, s, *p); break;
      default: hipLaunchKernelGGL(conv_few_kernel<4>, grid, dim3(256), 0, s, *p); break;
    }
    OMNI_CHECK_LAUNCH();
    return norm_after();
  }
  if (p->Cout % 96 == 0) {
    hipLaunchKernelGGL(conv2d_kernel<3>, dim3((unsigned)((M + CBM - 1) / CBM), p->Cout / 96), dim3(256), 0, s, *p);
  } else {
    hipLaunchKernelGGL(conv2d_kernel<1>, dim3((unsigned)((M + CBM - 1) / CBM), (p->Cout + 31) / 32), dim3(256), 0, s,
                       *p);
  }
  OMNI_CHECK_LAUNCH();
  return norm_after();
}

extern "C" int omni_vae_rmsnorm_silu(const omni_bf16* x, omni_bf16* y, int64_t rows, int32_t C, const omni_bf16* gamma,
                                     int32_t silu, omni_stream stream) {
  if (!x || !y || !gamma || rows <= 0 || C <= 0) return OMNI_ERR_BAD_ARG;
  if (C % 8 || C > 512) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(x) || !omni_aligned16(y) || !omni_aligned16(gamma)) return OMNI_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int lanes = C / 8;
#define OMNI_VN(L)                                                                                         \
  hipLaunchKernelGGL(vae_rmsnorm_kernel<L>, dim3((unsigned)((rows + (256 / L) - 1) / (256 / L))), dim3(256), 0, s, x, \
                     y, rows, C, gamma, silu)
  if (lanes <= 2) OMNI_VN(2);
  else if (lanes <= 16) OMNI_VN(16);
  else if (lanes <= 32) OMNI_VN(32);
  else OMNI_VN(64);
#undef OMNI_VN
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

extern "C" int omni_softmax_rows(omni_bf16* s, int64_t ld, int64_t rows, int32_t cols, float scale,
                                 omni_stream stream) {
  if (!s || rows <= 0 || cols <= 0) return OMNI_ERR_BAD_ARG;
  if (cols % 8) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(s) || (ld % 8)) return OMNI_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cols <= 64 * 256) hipLaunchKernelGGL(softmax_rows_reg_kernel<256>, dim3((unsigned)rows), dim3(256), 0, st, s, ld, cols, scale);
  else if (cols <= 64 * 1024) hipLaunchKernelGGL(softmax_rows_reg_kernel<1024>, dim3((unsigned)rows), dim3(1024), 0, st, s, ld, cols, scale);
  else hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, ld, cols, scale);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

extern "C" int omni_vae_attention(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out, int32_t B,
                                  int32_t tokens, int32_t C, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale,
                                  omni_stream stream) {
  if (!q || !k || !v || !out || B <= 0 || tokens <= 0) return OMNI_ERR_BAD_ARG;
  if (C != VA_DH || B > 65535) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(q) || !omni_aligned16(k) || !omni_aligned16(v) || (reinterpret_cast<uintptr_t>(out) & 7) || (ldq % 8) ||
      (ldk % 8) || (ldv % 8) || (ldo % 4))
    return OMNI_ERR_ALIGN;
  if ((int64_t)tokens * ldk * 2 >= (1ll << 32) || (int64_t)tokens * ldv * 2 >= (1ll << 32)) return OMNI_ERR_UNSUPPORTED;   // 32-bit offsets
  static std::atomic<uint64_t> attr_done{0};
  OMNI_TRY_STATUS(omni_once_per_device(attr_done, [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(vae_attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, VA_LDS) ==
           hipSuccess;
  }));
  hipLaunchKernelGGL(vae_attn_fwd_kernel, dim3((unsigned)((tokens + 32 * VA_NW - 1) / (32 * VA_NW)), B), dim3(VA_NW * 64), VA_LDS,
                     static_cast<hipStream_t>(stream), q, k, v, out, ldq, ldk, ldv, ldo, tokens, scale * 1.4426950408889634f);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

extern "C" int omni_vae_upsample2x_bordered(const omni_bf16* x, omni_bf16* y, int32_t B, int32_t H, int32_t W, int32_t C,
                                            omni_stream stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return OMNI_ERR_BAD_ARG;
  if (C % 8) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(x) || !omni_aligned16(y)) return OMNI_ERR_ALIGN;
  const int64_t total = (int64_t)B * (2 * H + 2) * (2 * W + 2) * (C / 8);
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(upsample2x_bordered_kernel, dim3((unsigned)(blocks < 65536 * 4 ? blocks : 65536 * 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, B, H, W, C);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}
