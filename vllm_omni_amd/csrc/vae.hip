// VAE-decode kernels for gfx950 (AutoencoderKLQwenImage.decode for one frame).
//
//  * omni_vae_conv2d: 3x3 (pad 1) / 1x1 convolution as an implicit GEMM on v_mfma_f32_32x32x16_bf16.
//    Activations are NHWC bf16, weights [Cout][ky][kx][Cin] (the temporal slice [-1] of the reference's causal
//    Conv3d weights: for a single frame the two zero front-pad frames make the other two slices dead,
//    autoencoder_kl_qwenimage.py:69-84 — executing them, as the reference does, is 3x the MACs).
//    M = output pixels, N = Cout, K = ks*ks*Cin.  Workgroup tile 128 pixels x (32*NB) channels, 4 waves,
//    each wave 32 pixels x 32*NB channels; swapped MFMA operands so a lane owns one pixel and 4 consecutive
//    channels per register quad (8-byte NHWC stores).  The im2col gather (with optional fused nearest-exact
//    x2 upsample: source = dst >> 1, QwenImageUpsample :112-124) happens in the global->register stage;
//    bias, residual add and clamp are fused in the epilogue.
//  * omni_vae_rmsnorm_silu: y = silu(x / max(||x||_2, 1e-12) * sqrt(C) * gamma) per pixel (QwenImageRMS_norm
//    :108-109 + SiLU), 16 B per lane.
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
        const int uy = a_oy[i] + ky - pad, ux = a_ox[i] + kx - pad;
        if (uy >= 0 && uy < Hout && ux >= 0 && ux < Wout) {
          const int iy = P.upsample2x ? (uy >> 1) : uy, ix = P.upsample2x ? (ux >> 1) : ux;
          v = *reinterpret_cast<const u32x4_t*>(x + (((int64_t)a_b[i] * P.Hin + iy) * P.Win + ix) * P.Cin + ci);
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

// in-place softmax over rows of `cols` bf16 scores: p = exp((s - max) * scale) / sum.  One workgroup per row.
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

extern "C" int omni_vae_conv2d(const omni_conv_params* p, omni_stream stream) {
  if (!p || !p->x || !p->w || !p->y || p->B <= 0 || p->Hin <= 0 || p->Win <= 0 || p->Cin <= 0 || p->Cout <= 0)
    return OMNI_ERR_BAD_ARG;
  if ((p->ksize != 1 && p->ksize != 3) || p->Cin % 8) return OMNI_ERR_UNSUPPORTED;
  if (p->gamma) return OMNI_ERR_UNSUPPORTED;  // fused norm prologue: not built yet (use omni_vae_rmsnorm_silu)
  if (p->downsample2x && (p->upsample2x || p->ksize != 3 || (p->Hin & 1) || (p->Win & 1))) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(p->x) || !omni_aligned16(p->w)) return OMNI_ERR_ALIGN;
  const int Hout = p->upsample2x ? 2 * p->Hin : (p->downsample2x ? p->Hin / 2 : p->Hin);
  const int Wout = p->upsample2x ? 2 * p->Win : (p->downsample2x ? p->Win / 2 : p->Win);
  const int64_t M = (int64_t)p->B * Hout * Wout;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (p->Cout % 96 == 0) {
    hipLaunchKernelGGL(conv2d_kernel<3>, dim3((unsigned)((M + CBM - 1) / CBM), p->Cout / 96), dim3(256), 0, s, *p);
  } else {
    hipLaunchKernelGGL(conv2d_kernel<1>, dim3((unsigned)((M + CBM - 1) / CBM), (p->Cout + 31) / 32), dim3(256), 0, s,
                       *p);
  }
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
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
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, static_cast<hipStream_t>(stream), s, ld,
                     cols, scale);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}
