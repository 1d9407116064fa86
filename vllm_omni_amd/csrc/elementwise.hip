// HBM-bound kernels of the DiT path for gfx950: AdaLN-modulate, RMSNorm, per-head RMSNorm+RoPE, RoPE,
// small-batch weight-streaming linear (modulation / timestep GEMVs), timestep sinusoid, CFG+Euler step.
// All of them move 16 bytes per lane per access (8 bf16) and do their arithmetic in fp32 with one rounding.
// Roofline: HBM.  Algorithmic bytes are stated per kernel.
#include "common.h"

namespace {

OMNI_DEVINL void unpack8(const u32x4_t& w, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf16_lo(w[i]);
    f[2 * i + 1] = bf16_hi(w[i]);
  }
}
OMNI_DEVINL u32x4_t pack8(const float* f) {
  u32x4_t w;
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
  return w;
}

OMNI_DEVINL uint32_t cvt_pk_fp8x4(float a, float b, float c, float d) {
  uint32_t r = 0;
  asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2\n\tv_cvt_pk_fp8_f32 %0, %3, %4 op_sel:[0,0,1]" : "+v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
  return r;
}

// ------------------------------------------------------------------------------------------------
// AdaLN-modulate / RMSNorm: one wave per row, the row lives in registers (NCH chunks of 512 elements).
// bytes/row = 2*D (read) + 2*D (write) + modulation vectors (L2-resident).
// MODE 0: y = LN(x)*(1+scale)+shift    MODE 1: y = x*rsqrt(mean(x^2)+eps)*w
// ------------------------------------------------------------------------------------------------
// PAIR (round 6): ONE launch over two row groups that share D / ld / eps / the modulation stride — the image stream's rows and,
// behind them, the text stream's (`g2`).  A DiT block runs AdaLN on both streams back to back; at small batches (one 256^2 CFG
// pair: 512 + 128 rows) each launch is ~5 us of latency plus its boundary, four times per block.  One wave still owns one row:
// the group is a wave-uniform select of the pointers.
struct RowNormGroup {
  const uint16_t* x; uint16_t* y; const uint16_t* scale; const uint16_t* shift; const int32_t* row_item_map;
  uint8_t* y8; float* y8_scale; int rows, rows_per_item, y_k32_rows, y8_rows;
};
template <int NCH, int MODE, bool PAIR = false>
__global__ __launch_bounds__(256) void rownorm_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                      uint16_t* __restrict__ y, int64_t ldy, int rows, int D,
                                                      const uint16_t* __restrict__ scale_or_w,
                                                      const uint16_t* __restrict__ shift, int64_t item_stride,
                                                      const int32_t* __restrict__ row_item_map, int rows_per_item,
                                                      float eps, int y_k32_rows, uint8_t* __restrict__ y8,
                                                      int y8_rows, float* __restrict__ y8_scale, const RowNormGroup g2) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if constexpr (PAIR) {
    row = __builtin_amdgcn_readfirstlane(row);
    if (row >= rows) {                                 // a row of the second group (uniform over the wave)
      row -= rows;
      x = g2.x; y = g2.y; scale_or_w = g2.scale; shift = g2.shift; row_item_map = g2.row_item_map;
      y8 = g2.y8; y8_scale = g2.y8_scale; rows = g2.rows; rows_per_item = g2.rows_per_item; y_k32_rows = g2.y_k32_rows;
      y8_rows = g2.y8_rows;
    }
  }
  if (row >= rows) return;
  const uint16_t* xr = x + (int64_t)row * ldx;
  // ALL of the row's loads are issued before the first value is used (round 4: with the load inside `if (e < D) { load; use }`
  // hipcc waited vmcnt(0) after every chunk - six serial HBM round trips per row and ONE KiB in flight per wave; the kernel ran
  // at 3.9 TB/s = exactly 16 waves x 1 KiB x 256 CUs per microsecond of latency).  Chunks past D read a clamped address and
  // are masked afterwards: no branch around a load.  The row stays PACKED (bf16 pairs, NCH x 4 registers) and is unpacked
  // in each pass, so that the modulation vectors fit beside it at 4 waves per SIMD.
  int item = 0;                                        // requested FIRST: the modulation loads below wait for it alone (a counted
  if (MODE == 0) item = row_item_map ? row_item_map[row] : row / rows_per_item;   // vmcnt), not for the row
  u32x4_t raw[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) raw[c] = *reinterpret_cast<const u32x4_t*>(xr + min((c * 64 + lane) * 8, D - 8));
  __builtin_amdgcn_sched_barrier(0);                   // keep the first USE of `item` (and its wait) behind the row's loads
  const int64_t moff = (int64_t)item * item_stride;
  u32x4_t rsc[NCH], rsh[NCH];                          // the row's modulation vectors (L2-resident), requested with the row
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = min((c * 64 + lane) * 8, D - 8);
    rsc[c] = *reinterpret_cast<const u32x4_t*>(scale_or_w + moff + e);
    if (MODE == 0) rsh[c] = *reinterpret_cast<const u32x4_t*>(shift + moff + e);
  }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if ((c * 64 + lane) * 8 < D) {
      float v[8];
      unpack8(raw[c], v);
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += (MODE == 0) ? v[i] : v[i] * v[i];
    }
  }
  sum = wave_sum<64>(sum);
  float mean = 0.f, rstd;
  if (MODE == 0) {
    mean = sum / D;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if ((c * 64 + lane) * 8 < D) {
        float v[8];
        unpack8(raw[c], v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d = v[i] - mean;
          var += d * d;
        }
      }
    }
    var = wave_sum<64>(var) / D;
    rstd = rsqrtf(var + eps);
  } else {
    rstd = rsqrtf(sum / D + eps);
  }
  uint16_t* yr = y + (int64_t)row * ldy;
  float amax = 0.0f;                                   // fp8 output: per-row amax of the bf16-ROUNDED result
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < D) {
      float v[8], sc[8], sh[8], o[8];
      unpack8(raw[c], v);
      unpack8(rsc[c], sc);
      if (MODE == 0) {
        unpack8(rsh[c], sh);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (v[i] - mean) * rstd * (1.0f + sc[i]) + sh[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = v[i] * rstd * sc[i];
      }
      const u32x4_t pk = pack8(o);
      if (y) {
        // K32-blocked output: the 8 elements stay one 16-B granule, at ((e/32) * R + row) * 32 + e%32
        uint16_t* dst = y_k32_rows ? y + ((int64_t)(e >> 5) * y_k32_rows + row) * 32 + (e & 31) : yr + e;
        *reinterpret_cast<u32x4_t*>(dst) = pk;
      }
      if (MODE == 0 && y8) {                           // keep the bf16-ROUNDED values (in the row's registers): the same numbers
        raw[c] = pk;                                   // omni_quantize_fp8_rows would read back from y: fused == unfused bit for bit
        float r[8];
        unpack8(pk, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(r[i]));
      }
    }
  }
  if (MODE == 0 && y8) {
    amax = wave_max<64>(amax);
    const float q = fmaxf(amax, 1e-12f) * (1.0f / 448.0f), inv = 1.0f / q;
    if (lane == 0) y8_scale[row] = q;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e = (c * 64 + lane) * 8;
      if (e < D) {
        float r[8];
        unpack8(raw[c], r);
        u32x2_t o8;
        o8[0] = cvt_pk_fp8x4(r[0] * inv, r[1] * inv, r[2] * inv, r[3] * inv);
        o8[1] = cvt_pk_fp8x4(r[4] * inv, r[5] * inv, r[6] * inv, r[7] * inv);
        *reinterpret_cast<u32x2_t*>(y8 + ((int64_t)(e >> 6) * y8_rows + row) * 64 + (e & 63)) = o8;
      }
    }
  }
}

template <int MODE>
int launch_rownorm(const omni_bf16* x, int64_t ldx, omni_bf16* y, int64_t ldy, int rows, int D, const omni_bf16* a,
                   const omni_bf16* b, int64_t stride, const int32_t* map, int rpi, float eps, hipStream_t s,
                   int y_k32_rows = 0, uint8_t* y8 = nullptr, int y8_rows = 0, float* y8_scale = nullptr) {
  const int nch = (D + 511) / 512;
  const dim3 grid((rows + 3) / 4), block(256);
#define OMNI_RN(N)                                                                                            \
  hipLaunchKernelGGL((rownorm_kernel<N, MODE>), grid, block, 0, s, x, ldx, y, ldy, rows, D, a, b, stride, map, \
                     rpi, eps, y_k32_rows, y8, y8_rows, y8_scale, RowNormGroup{})
  if (nch <= 1) OMNI_RN(1);
  else if (nch <= 2) OMNI_RN(2);
  else if (nch <= 4) OMNI_RN(4);
  else if (nch <= 6) OMNI_RN(6);
  else if (nch <= 8) OMNI_RN(8);
  else if (nch <= 16) OMNI_RN(16);
  else return OMNI_ERR_UNSUPPORTED;
#undef OMNI_RN
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

// AdaLN over two row groups in one launch (see rownorm_kernel PAIR)
int launch_adaln_pair(const omni_adaln_stream& a, const omni_adaln_stream& b, int D, int64_t stride, float eps, hipStream_t s) {
  const int nch = (D + 511) / 512;
  const dim3 grid((a.rows + b.rows + 3) / 4), block(256);
  const RowNormGroup g2 = {b.x, b.y, b.scale, b.shift, b.row_item_map, b.y8, b.y8_scale, b.rows, b.rows_per_item,
                           b.y_k32_rows, b.y8_rows};
#define OMNI_RN(N)                                                                                                      \
  hipLaunchKernelGGL((rownorm_kernel<N, 0, true>), grid, block, 0, s, a.x, (int64_t)D, a.y, (int64_t)D, a.rows, D, a.scale,   \
                     a.shift, stride, a.row_item_map, a.rows_per_item, eps, a.y_k32_rows, a.y8, a.y8_rows, a.y8_scale, g2)
  if (nch <= 1) OMNI_RN(1);
  else if (nch <= 2) OMNI_RN(2);
  else if (nch <= 4) OMNI_RN(4);
  else if (nch <= 6) OMNI_RN(6);
  else if (nch <= 8) OMNI_RN(8);
  else if (nch <= 16) OMNI_RN(16);
  else return OMNI_ERR_UNSUPPORTED;
#undef OMNI_RN
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

// ------------------------------------------------------------------------------------------------
// Split-K finish + gated residual + AdaLN in one pass (ABI v13, omni_splitk_finish_adaln_pair).  At one 256^2 CFG pair (512 + 128
// rows) the out-projection and the MLP down-projection run 6-way K-split; their finish kernel (47 MB of fp32 partials -> the
// residual stream, ~15 us) is followed at once by the AdaLN of that stream (~5 us): three launches, each latency-bound, and the
// residual row written only to be read back.  Here ONE wave owns one row of either stream:
//   c = bf16(sum_s partial[s][row][:] + bias)   the sum in split order from 0.0f, exactly gemm_epilogue_lds_impl<FROM_PARTIALS>
//   h = bf16(fma(gate, c, res))                 (hipcc contracts the epilogue's `res + gate * c` to an fma: the same bits)
//   y = AdaLN(h)                                rownorm_kernel<MODE 0>'s arithmetic on the ROUNDED row, operation for operation:
//                                               sequential sums, IEEE divisions, var += d * d as a mul and an add (hipcc does not
//                                               contract that one: the products are formed pairwise, v_pk_mul_f32), the output as
//                                               fma((v - mean) * rstd, 1 + scale, shift)
// with implicit contraction OFF in this function, so that what is an fma is written as one — tests/test_gpu_finish_adaln.py
// asserts bit equality with the two-kernel path for both streams, row-major and K32-blocked.
// Loads: the partials of chunk c + 2 (NS x 32 B per lane), its residual, gate and bias pieces are requested while chunk c is
// reduced (two chunks in flight: 24 x 16 B per lane at NS = 6 — 640 waves x 64 lanes x 384 B = 15.7 MB in flight).
// ------------------------------------------------------------------------------------------------
struct FinishGroup {
  const float* ws;             // partials of this group's first row in split 0
  const uint16_t* bias;        // nullable
  uint16_t* hidden;            // [rows, D], read (residual) and written (result) in place
  const uint16_t* gate; const uint16_t* scale; const uint16_t* shift;
  const int32_t* row_item_map;
  uint16_t* y;
  int rows, rows_per_item, y_k32_rows;
};
template <int NCH, int NS>
__global__ __launch_bounds__(64) void splitk_finish_adaln_kernel(const FinishGroup g1, const FinishGroup g2, int64_t split_stride,
                                                                 int D, int64_t item_stride, float eps) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x;
  int row = blockIdx.x;                                // one wave = one workgroup = one row (uniform)
  const float* ws = g1.ws; const uint16_t* bias = g1.bias; uint16_t* hidden = g1.hidden;
  const uint16_t *gate = g1.gate, *scale = g1.scale, *shift = g1.shift;
  const int32_t* row_item_map = g1.row_item_map; uint16_t* y = g1.y;
  int rows = g1.rows, rows_per_item = g1.rows_per_item, y_k32_rows = g1.y_k32_rows;
  if (row >= g1.rows) {                                // a row of the second group: field-wise (a struct select goes to scratch)
    row -= g1.rows;
    ws = g2.ws; bias = g2.bias; hidden = g2.hidden; gate = g2.gate; scale = g2.scale; shift = g2.shift;
    row_item_map = g2.row_item_map; y = g2.y; rows = g2.rows; rows_per_item = g2.rows_per_item; y_k32_rows = g2.y_k32_rows;
  }
  if (row >= rows) return;
  const int item = row_item_map ? row_item_map[row] : row / rows_per_item;    // requested first: only the gate / scale / shift
  const float* wr = ws + (int64_t)row * D;                                    // loads wait for it
  uint16_t* hr = hidden + (int64_t)row * D;
  f32x4_t pl[2][NS], ph[2][NS];
  u32x4_t rr[2], bb[2], gg[2];
  auto issue_partials = [&](int c, int buf) {
    const int e = min((c * 64 + lane) * 8, D - 8);     // chunks past D read a clamped address and are masked afterwards
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float* q = wr + (int64_t)s * split_stride + e;
      pl[buf][s] = *reinterpret_cast<const f32x4_t*>(q);
      ph[buf][s] = *reinterpret_cast<const f32x4_t*>(q + 4);
    }
    rr[buf] = *reinterpret_cast<const u32x4_t*>(hr + e);
    bb[buf] = u32x4_t{0u, 0u, 0u, 0u};
    if (bias) bb[buf] = *reinterpret_cast<const u32x4_t*>(bias + e);
  };
  issue_partials(0, 0);
  if (NCH > 1) issue_partials(1, 1);
  __builtin_amdgcn_sched_barrier(0);                   // keep the first USE of `item` (and its wait) behind those loads
  const int64_t moff = (int64_t)item * item_stride;
  auto issue_gate = [&](int c, int buf) {
    gg[buf] = *reinterpret_cast<const u32x4_t*>(gate + moff + min((c * 64 + lane) * 8, D - 8));
  };
  issue_gate(0, 0);
  if (NCH > 1) issue_gate(1, 1);
  u32x4_t rsc[NCH], rsh[NCH];                          // the AdaLN vectors of the row's item (L2-resident)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = min((c * 64 + lane) * 8, D - 8);
    rsc[c] = *reinterpret_cast<const u32x4_t*>(scale + moff + e);
    rsh[c] = *reinterpret_cast<const u32x4_t*>(shift + moff + e);
  }
  u32x4_t raw[NCH];                                    // the new residual row, PACKED (as rownorm_kernel keeps it)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int buf = c & 1;
    f32x4_t lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
    for (int s = 0; s < NS; ++s) { lo += pl[buf][s]; hi += ph[buf][s]; }
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if (bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[2 * i] += bf16_lo(bb[buf][i]); v[2 * i + 1] += bf16_hi(bb[buf][i]); }
    }
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t cw = pack_bf16x2(v[2 * i], v[2 * i + 1]);       // the GEMM output, rounded where the fused epilogue rounds it
      o[i] = pack_bf16x2(__builtin_fmaf(bf16_lo(gg[buf][i]), bf16_lo(cw), bf16_lo(rr[buf][i])),
                         __builtin_fmaf(bf16_hi(gg[buf][i]), bf16_hi(cw), bf16_hi(rr[buf][i])));
    }
    raw[c] = o;
    if ((c * 64 + lane) * 8 < D) *reinterpret_cast<u32x4_t*>(hr + (c * 64 + lane) * 8) = o;
    if (c + 2 < NCH) { issue_partials(c + 2, buf); issue_gate(c + 2, buf); }
  }
  // ---- AdaLN of the row: rownorm_kernel<NCH, 0>'s passes
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if ((c * 64 + lane) * 8 < D) {
      float v[8];
      unpack8(raw[c], v);
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += v[i];
    }
  }
  sum = wave_sum<64>(sum);
  const float mean = sum / D;
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if ((c * 64 + lane) * 8 < D) {
      float v[8];
      unpack8(raw[c], v);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = v[i] - mean;
        const float dd = d * d;
        var += dd;
      }
    }
  }
  var = wave_sum<64>(var) / D;
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * 64 + lane) * 8;
    if (e < D) {
      float v[8], sc[8], sh[8], o[8];
      unpack8(raw[c], v);
      unpack8(rsc[c], sc);
      unpack8(rsh[c], sh);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf((v[i] - mean) * rstd, 1.0f + sc[i], sh[i]);
      uint16_t* dst = y_k32_rows ? y + ((int64_t)(e >> 5) * y_k32_rows + row) * 32 + (e & 31) : y + (int64_t)row * D + e;
      *reinterpret_cast<u32x4_t*>(dst) = pack8(o);
    }
  }
}

int launch_finish_adaln(const float* ws, int nsplit, int64_t ws_rows, const omni_finish_adaln_stream& a,
                        const omni_finish_adaln_stream& b, int D, int64_t stride, float eps, hipStream_t s) {
  auto grp = [&](const omni_finish_adaln_stream& g) {
    return FinishGroup{ws + (int64_t)g.ws_row0 * D, g.bias, g.hidden, g.gate, g.scale, g.shift, g.row_item_map, g.y,
                       g.rows, g.rows_per_item, g.y_k32_rows};
  };
  const FinishGroup g1 = grp(a), g2 = grp(b);
  const dim3 grid(a.rows + b.rows), block(64);
  const int64_t split_stride = ws_rows * D;
  const int nch = (D + 511) / 512;
#define OMNI_FA(N, S) hipLaunchKernelGGL((splitk_finish_adaln_kernel<N, S>), grid, block, 0, s, g1, g2, split_stride, D, stride, eps)
#define OMNI_FA_NS(N)                    \
  switch (nsplit) {                      \
    case 2: OMNI_FA(N, 2); break;        \
    case 3: OMNI_FA(N, 3); break;        \
    case 4: OMNI_FA(N, 4); break;        \
    case 6: OMNI_FA(N, 6); break;        \
    case 8: OMNI_FA(N, 8); break;        \
    default: return OMNI_ERR_UNSUPPORTED; \
  }
  if (nch <= 2) { OMNI_FA_NS(2) }
  else if (nch <= 4) { OMNI_FA_NS(4) }
  else if (nch <= 6) { OMNI_FA_NS(6) }
  else if (nch <= 8) { OMNI_FA_NS(8) }
  else return OMNI_ERR_UNSUPPORTED;
#undef OMNI_FA_NS
#undef OMNI_FA
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

// ------------------------------------------------------------------------------------------------
// Per-head RMSNorm(128) + interleaved RoPE, in place.  16 lanes x 8 elements = one (row, head);
// a wave handles 4 (row, head) pairs per instruction.  bytes = 2 * rows*H*128*2 (read + write).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(uint16_t* __restrict__ x, int64_t ldx, int rows, int H,
                                                           const uint16_t* __restrict__ w_img,
                                                           const uint16_t* __restrict__ w_txt,
                                                           const uint16_t* __restrict__ cos_tab,
                                                           const uint16_t* __restrict__ sin_tab,
                                                           const int32_t* __restrict__ row_pos, int txt_pos_end,
                                                           float eps) {
  const int sub = threadIdx.x & 15;
  const int64_t unit = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);  // (row, head) pair index
  if (unit >= (int64_t)rows * H) return;
  const int row = (int)(unit / H), head = (int)(unit - (int64_t)row * H);
  uint16_t* p = x + (int64_t)row * ldx + head * 128 + sub * 8;
  float f[8], w[8], o[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(p), f);
  const int pos = row_pos[row];
  unpack8(*reinterpret_cast<const u32x4_t*>((pos < txt_pos_end ? w_txt : w_img) + sub * 8), w);
  const u32x2_t cw = *reinterpret_cast<const u32x2_t*>(cos_tab + (int64_t)pos * 64 + sub * 4);
  const u32x2_t sw = *reinterpret_cast<const u32x2_t*>(sin_tab + (int64_t)pos * 64 + sub * 4);
  const float c[4] = {bf16_lo(cw[0]), bf16_hi(cw[0]), bf16_lo(cw[1]), bf16_hi(cw[1])};
  const float s[4] = {bf16_lo(sw[0]), bf16_hi(sw[0]), bf16_lo(sw[1]), bf16_hi(sw[1])};
  qk_norm_rope_lane(f, w, c, s, eps, o);
  *reinterpret_cast<u32x4_t*>(p) = pack8(o);
}

// Standalone interleaved RoPE on [B,S,H,dh]; one lane per 8 elements.
__global__ __launch_bounds__(256) void rope_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                   int64_t total_chunks, int S, int H, int dh,
                                                   const uint16_t* __restrict__ cos_tab,
                                                   const uint16_t* __restrict__ sin_tab) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= total_chunks) return;
  const int cpr = dh / 8;  // chunks per head row
  const int sub = (int)(id % cpr);
  const int64_t tok_head = id / cpr;
  const int s_idx = (int)((tok_head / H) % S);
  float f[8], o[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(x + id * 8), f);
  const u32x2_t cw = *reinterpret_cast<const u32x2_t*>(cos_tab + (int64_t)s_idx * (dh / 2) + sub * 4);
  const u32x2_t sw = *reinterpret_cast<const u32x2_t*>(sin_tab + (int64_t)s_idx * (dh / 2) + sub * 4);
  const float c[4] = {bf16_lo(cw[0]), bf16_hi(cw[0]), bf16_lo(cw[1]), bf16_hi(cw[1])};
  const float s[4] = {bf16_lo(sw[0]), bf16_hi(sw[0]), bf16_lo(sw[1]), bf16_hi(sw[1])};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = f[2 * i] * c[i] - f[2 * i + 1] * s[i];
    o[2 * i + 1] = f[2 * i + 1] * c[i] + f[2 * i] * s[i];
  }
  *reinterpret_cast<u32x4_t*>(y + id * 8) = pack8(o);
}

// ------------------------------------------------------------------------------------------------
// Small-batch linear (weight streaming):  y[b,n] = act_out(sum_k act_in(x[b,k]) W[n,k] + bias[n]).
// act_in(x) is staged once per workgroup into LDS as fp32; each wave then streams whole weight rows
// (16 B per lane per load, ROWS_PER_ITER rows in flight) and reduces across the wave.
// bytes = N*K*2 (W, read once) — the 13.6 GB/forward modulation stream of the DiT.
// ------------------------------------------------------------------------------------------------
// MAXC = 16-byte pieces per lane and weight row (K <= 512 * MAXC): a template parameter since round 6's third session — with the
// fixed 8 a K = 3072 row (6 pieces) issued two more loads of its last piece per row, a quarter of the kernel's load instructions.
template <int NB, int MAXC>
__global__ __launch_bounds__(256) void linear_smallbatch_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                const uint16_t* __restrict__ W,
                                                                const uint16_t* __restrict__ bias, int64_t N, int K,
                                                                uint16_t* __restrict__ y, int64_t ldy, int act_in,
                                                                int act_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);  // [NB][K]
  // staging: 16-byte pieces (8 activations) per thread and trip when the rows allow it.  The scalar form (one bf16 per thread
  // and trip: 48 dependent trips at NB = 4, K = 3072) cost as much as streaming the block's weight rows — the block count is
  // now sized so that a block streams at least three times the bytes it stages (host side).
  if ((ldx % 8) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const int nch = K / 8;
    for (int i = threadIdx.x; i < NB * nch; i += 256) {
      const int b = i / nch, c = i - b * nch;
      const u32x4_t w = *reinterpret_cast<const u32x4_t*>(x + (int64_t)b * ldx + c * 8);
      float f[8];
      unpack8(w, f);
      if (act_in == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
      }
      float* dst = xs + b * K + c * 8;
      *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{f[0], f[1], f[2], f[3]};
      *reinterpret_cast<f32x4_t*>(dst + 4) = f32x4_t{f[4], f[5], f[6], f[7]};
    }
  } else {
    for (int i = threadIdx.x; i < NB * K; i += 256) {
      const int b = i / K, kk = i - b * K;
      float v = bf16_bits_to_f32(x[(int64_t)b * ldx + kk]);
      if (act_in == 1) v = silu_f(v);
      xs[i] = v;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int nchunk = K / 8;  // 16-B chunks per row
  constexpr int R = 2;       // weight rows in flight per wave
  // Round 6, third session: with FIVE or more staged rows (96 KB of LDS at 8: one workgroup per CU, four waves) the NEXT trip's
  // weight rows are requested before this trip's arithmetic — the table pass of 8 rows: 7.9 -> 5.2 ms, 6 rows 4.2 -> 3.8.  Up to four
  // rows (three workgroups per CU) the same change costs 12 .. 18 % (2.5 -> 3.0 ms at one row): the second row pair's registers buy
  // nothing there, the loads of twelve resident waves already cover each other (profiles/r06e_table_pass.txt).
  constexpr bool AHEAD = NB >= 5;
  u32x4_t w[R][MAXC], wn[R][AHEAD ? MAXC : 1];
  auto load_rows = [&](int64_t n0, auto& dst) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint16_t* wrow = W + min(n0 + r, N - 1) * K;
#pragma unroll
      for (int j = 0; j < MAXC; ++j)
        dst[r][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wrow + min(lane + 64 * j, nchunk - 1) * 8));
    }
  };
  if (AHEAD && wave_global * R < N) load_rows(wave_global * R, w);
  for (int64_t n0 = wave_global * R; n0 < N; n0 += nwaves * R) {
    const bool more = AHEAD && n0 + nwaves * R < N;      // uniform over the wave
    if constexpr (AHEAD) {
      if (more) load_rows(n0 + nwaves * R, wn);
    } else {
      load_rows(n0, w);
    }
    float acc[R][NB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
    // ALL of the two rows' 16-byte pieces (K <= 4096: at most 8 per lane and row) are requested before the first one is used:
    // with a load per loop trip (round 1's form) a wave had 2 KB in flight and the table pass of a 4-step request (NB = 4: 48 KB
    // of staged activations, three workgroups per CU) streamed the modulation weights at 1.9 TB/s (profiles/r06_first_profiles_
    // step_shapes.txt: 59 us per 113 MB matrix).  Pieces past the row end re-read the last piece (unconditional loads can be
    // hoisted; a load under `if (c < nchunk)` is waited for on the spot) and are skipped in the arithmetic.
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
      const int c = lane + 64 * j;
      if (c >= nchunk) break;
      float xv[NB][8];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(xs + b * K + c * 8);
        const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(xs + b * K + c * 8 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xv[b][i] = a0[i];
          xv[b][4 + i] = a1[i];
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float wf[8];
        unpack8(w[r][j], wf);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[r][b] += wf[i] * xv[b][i];
      }
    }
    // R x NB wave reductions per iteration: VALU-only (DPP), the total lands in lane 63 (round 6, third session: the butterfly form
    // was 48 ds_bpermute round trips per iteration at NB = 4, in the LDS pipe that also serves the activation reads)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[r][b] = wave_sum_to_lane63(acc[r][b]);
    if (lane == 63) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t n = n0 + r;
        if (n < N) {
          const float bv = bias ? bf16_bits_to_f32(bias[n]) : 0.f;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            float v = acc[r][b] + bv;
            if (act_out == 1) v = silu_f(v);
            y[(int64_t)b * ldy + n] = f32_to_bf16_bits(v);
          }
        }
      }
    }
    if constexpr (AHEAD) {
      if (more) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int j = 0; j < MAXC; ++j) w[r][j] = wn[r][j];
      }
    }
  }
}

// Timesteps(dim, flip_sin_to_cos=True, shift 0, scale): out[b] = [cos(t*scale*f_i) | sin(t*scale*f_i)]
__global__ void timestep_sinusoid_kernel(const float* __restrict__ t, int B, int dim, float scale,
                                         uint16_t* __restrict__ out) {
  const int half = dim / 2;
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * half) return;
  const int b = id / half, i = id - b * half;
  const float freq = expf(-9.210340371976184f * (float)i / (float)half);  // ln(10000)
  const float arg = scale * (t[b] * freq);
  out[(int64_t)b * dim + i] = f32_to_bf16_bits(cosf(arg));
  out[(int64_t)b * dim + half + i] = f32_to_bf16_bits(sinf(arg));
}

// ------------------------------------------------------------------------------------------------
// CFG combine + norm rescale + Euler update on [rows, 64] bf16: 8 lanes per row.
// bytes/row = 3 reads + 1 write of 128 B.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cfg_euler_kernel(const uint16_t* __restrict__ pos,
                                                        const uint16_t* __restrict__ neg,
                                                        uint16_t* __restrict__ lat, int rows, float cfg_scale,
                                                        const float* __restrict__ dt, int dt_rows_per_item, int normalize) {
  const int sub = threadIdx.x & 7;
  const int row = blockIdx.x * 32 + (threadIdx.x >> 3);
  if (row >= rows) return;
  const int64_t off = (int64_t)row * 64 + sub * 8;
  float p[8], x[8], pred[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(pos + off), p);
  unpack8(*reinterpret_cast<const u32x4_t*>(lat + off), x);
  if (neg) {
    float n[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(neg + off), n);
    float pp = 0.f, cc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      pred[i] = n[i] + cfg_scale * (p[i] - n[i]);
      pp += p[i] * p[i];
      cc += pred[i] * pred[i];
    }
    pp = wave_sum<8>(pp);
    cc = wave_sum<8>(cc);
    const float r = normalize ? sqrtf(pp) / sqrtf(cc) : 1.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) pred[i] *= r;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) pred[i] = p[i];
  }
  const float d = dt_rows_per_item > 0 ? dt[row / dt_rows_per_item] : dt[0];
  float o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = x[i] + d * pred[i];
  *reinterpret_cast<u32x4_t*>(lat + off) = pack8(o);
}

}  // namespace

namespace {
// dst[i] += src[i] on bf16 (one rounding of the sum): conditioning = timestep_emb + addition_t_emb of the Layered variant
__global__ __launch_bounds__(256) void add_bf16_kernel(uint16_t* __restrict__ dst, const uint16_t* __restrict__ src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = f32_to_bf16_bits(bf16_bits_to_f32(dst[i]) + bf16_bits_to_f32(src[i]));
}
__global__ __launch_bounds__(256) void silu_bf16_kernel(uint16_t* __restrict__ dst, const uint16_t* __restrict__ src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = f32_to_bf16_bits(silu_f(bf16_bits_to_f32(src[i])));
}
__global__ __launch_bounds__(256) void gather_i32_kernel(int32_t* __restrict__ dst, const int32_t* __restrict__ src,
                                                         const int32_t* __restrict__ idx, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
}  // namespace

int omni_internal_gather_i32(int32_t* dst, const int32_t* src, const int32_t* idx, int32_t n, void* stream) {
  if (!dst || !src || !idx || n <= 0) return OMNI_ERR_BAD_ARG;
  hipLaunchKernelGGL(gather_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), dst, src,
                     idx, n);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}


// ------------------------------------------------------------------------------------------------
// TeaCache on the device (reference vllm_omni/diffusion/cache/teacache/hook.py:170-217; see omni_teacache in the header).
// ------------------------------------------------------------------------------------------------
namespace {
// Partial sums of |mod - prev| and |prev| per item, and prev <- mod, in one pass.  The buffer is NSEG x n_items segments of
// `seg_elems` contiguous elements: (NSEG, W) = (D/32, 32) for the K32-blocked layout [D/32][rows][32], (1, D) for row-major.
// grid = (blocks per segment, n_items, NSEG)
__global__ __launch_bounds__(256) void teacache_reduce_kernel(const uint16_t* __restrict__ mod, uint16_t* __restrict__ prev,
                                                              float* __restrict__ scratch, int64_t seg_stride_item,
                                                              int64_t seg_stride_slab, int64_t seg_elems) {
  const int item = blockIdx.y;
  const int64_t base = (int64_t)blockIdx.z * seg_stride_slab + (int64_t)item * seg_stride_item;
  const int64_t nchunk = seg_elems / 8;
  float sd = 0.f, sp = 0.f;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * blockDim.x) {
    const u32x4_t m = *reinterpret_cast<const u32x4_t*>(mod + base + c * 8);
    const u32x4_t p = *reinterpret_cast<const u32x4_t*>(prev + base + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float m0 = bf16_lo(m[e]), m1 = bf16_hi(m[e]), p0 = bf16_lo(p[e]), p1 = bf16_hi(p[e]);
      // the reference subtracts two bf16 tensors (result rounded to bf16) before abs().mean()
      sd += fabsf(bf16_bits_to_f32(f32_to_bf16_bits(m0 - p0))) + fabsf(bf16_bits_to_f32(f32_to_bf16_bits(m1 - p1)));
      sp += fabsf(p0) + fabsf(p1);
    }
    *reinterpret_cast<u32x4_t*>(prev + base + c * 8) = m;
  }
  sd = wave_sum<64>(sd);
  sp = wave_sum<64>(sp);
  __shared__ float red[2][4];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = sd; red[1][wv] = sp; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(scratch + 2 * item, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(scratch + 2 * item + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// one workgroup: per-item decision, statistics, and the per-row-tile predicates of both streams
__global__ __launch_bounds__(256) void teacache_decide_kernel(omni_teacache tc, int n_items, int rows_per_item,
                                                              int n_img_rows, int n_txt_rows, float inv_count) {
  __shared__ int s_skip[64];
  const int i = threadIdx.x;
  if (i < n_items) {
    int skip = 0;
    const int cnt = tc.cnt[i];
    float acc = tc.acc_dist[i];
    if (cnt == 0) {
      acc = 0.f;                                           // first forward of a generation: always compute (:185-188)
    } else {
      // the reference forms this ratio from bf16 tensors: .abs().mean() -> bf16, (+ 1e-8) -> bf16, division -> bf16
      auto rb = [](float x) { return bf16_bits_to_f32(f32_to_bf16_bits(x)); };
      const float rel = rb(rb(tc.scratch[2 * i] * inv_count) / rb(rb(tc.scratch[2 * i + 1] * inv_count) + 1e-8f));
      float r = tc.coeff[0];
#pragma unroll
      for (int c = 1; c < 5; ++c) r = r * rel + tc.coeff[c];   // numpy.poly1d, highest power first
      acc += fabsf(r);
      if (acc < tc.rel_l1_thresh) skip = 1;
      else acc = 0.f;
    }
    tc.acc_dist[i] = acc;
    tc.cnt[i] = cnt + 1;
    tc.skip[i] = skip;
    tc.skip_total[i] += skip;
    tc.scratch[2 * i] = 0.f;
    tc.scratch[2 * i + 1] = 0.f;
    s_skip[i] = skip;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < (n_img_rows + 255) / 256; t += blockDim.x) {
    const int first = (t * 256) / rows_per_item, last = min(t * 256 + 255, n_img_rows - 1) / rows_per_item;
    int all = 1;
    for (int it = first; it <= last; ++it) all &= s_skip[it];
    tc.tile_skip_img[t] = all;
  }
  for (int t = threadIdx.x; t < (n_txt_rows + 255) / 256; t += blockDim.x) {
    const int r0 = t * 256, r1 = min(t * 256 + 255, n_txt_rows - 1);
    int all = 1;
    for (int it = 0; it < n_items; ++it)
      if (tc.txt_cu[it] <= r1 && tc.txt_cu[it + 1] > r0) all &= s_skip[it];
    tc.tile_skip_txt[t] = all;
  }
}

// after the block stack: skipped items take hidden_in + cached residual; computed items refresh the residual
__global__ __launch_bounds__(256) void teacache_post_kernel(uint16_t* __restrict__ hidden, const uint16_t* __restrict__ hin,
                                                            uint16_t* __restrict__ res, const int32_t* __restrict__ skip,
                                                            int64_t nchunk, int chunks_per_item) {
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (int64_t)gridDim.x * blockDim.x) {
    const int item = (int)(c / chunks_per_item);
    const u32x4_t a = *reinterpret_cast<const u32x4_t*>(hin + c * 8);
    u32x4_t o;
    if (skip[item]) {
      const u32x4_t r = *reinterpret_cast<const u32x4_t*>(res + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(bf16_lo(a[e]) + bf16_lo(r[e]), bf16_hi(a[e]) + bf16_hi(r[e]));
      *reinterpret_cast<u32x4_t*>(hidden + c * 8) = o;
    } else {
      const u32x4_t h = *reinterpret_cast<const u32x4_t*>(hidden + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(bf16_lo(h[e]) - bf16_lo(a[e]), bf16_hi(h[e]) - bf16_hi(a[e]));
      *reinterpret_cast<u32x4_t*>(res + c * 8) = o;
    }
  }
}

// ---- dynamic per-row fp8 (OCP e4m3) quantisation (omni_quantize_fp8_rows) ------------------------------------------------
// One wave per row.  A lane owns the 16-byte chunks c = lane, lane + 64, ... of the row (8 bf16 each), so a K32-blocked source
// (64-B slabs) and a row-major one are both read as whole 16-B pieces; the row stays in registers between the amax pass and
// the conversion.  Output chunk c (8 bytes) goes to slab c >> 3 of the K64-blocked destination.

template <int NCH>   // NCH = chunks per lane (K = NCH * 64 * 8 at most)
__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const uint16_t* __restrict__ x, int64_t ldx, int x_k32_rows,
                                                                int rows, int K, uint8_t* __restrict__ y8, int y_rows,
                                                                float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunks = K >> 3;
  u32x4_t v[NCH];
  float amax = 0.0f;
  // every load of the row is issued before the first value is used (see rownorm_kernel: a load inside `if (c < nchunks) { load;
  // use }` made hipcc wait for each chunk in turn); chunks past K read a clamped address and are masked to zero afterwards
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = min(lane + i * 64, nchunks - 1);
    const uint16_t* src = x_k32_rows ? x + ((int64_t)(c >> 2) * x_k32_rows + row) * 32 + (c & 3) * 8
                                     : x + (int64_t)row * ldx + c * 8;
    v[i] = *reinterpret_cast<const u32x4_t*>(src);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (lane + i * 64 >= nchunks) v[i] = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(bf16_lo(v[i][e])), fabsf(bf16_hi(v[i][e]))));
  }
  amax = wave_max<64>(amax);
  const float sc = fmaxf(amax, 1e-12f) * (1.0f / 448.0f);       // e4m3fn: largest finite 448
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nchunks) {
      u32x2_t o;
      o[0] = cvt_pk_fp8x4(bf16_lo(v[i][0]) * inv, bf16_hi(v[i][0]) * inv, bf16_lo(v[i][1]) * inv, bf16_hi(v[i][1]) * inv);
      o[1] = cvt_pk_fp8x4(bf16_lo(v[i][2]) * inv, bf16_hi(v[i][2]) * inv, bf16_lo(v[i][3]) * inv, bf16_hi(v[i][3]) * inv);
      *reinterpret_cast<u32x2_t*>(y8 + ((int64_t)(c >> 3) * y_rows + row) * 64 + (c & 7) * 8) = o;
    }
  }
}
}  // namespace

int omni_internal_teacache_decide(const omni_teacache* tc, const omni_bf16* mod, int32_t n_items, int32_t rows_per_item,
                                  int32_t n_img_rows, int32_t n_txt_rows, int32_t D, int32_t blocked, void* stream) {
  if (!tc || !mod || !tc->prev_mod || !tc->prev_res || !tc->acc_dist || !tc->cnt || !tc->skip || !tc->skip_total ||
      !tc->scratch || !tc->tile_skip_img || !tc->tile_skip_txt || !tc->txt_cu)
    return OMNI_ERR_BAD_ARG;
  if (n_items <= 0 || n_items > 64 || rows_per_item <= 0 || n_items * rows_per_item != n_img_rows || D % 32)
    return OMNI_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nseg = blocked ? D / 32 : 1, W = blocked ? 32 : D;
  const int64_t seg_elems = (int64_t)rows_per_item * W;
  int bx = (int)((seg_elems / 8 + 255) / 256);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(teacache_reduce_kernel, dim3(bx, n_items, nseg), dim3(256), 0, s, mod, tc->prev_mod, tc->scratch,
                     seg_elems, (int64_t)n_img_rows * W, seg_elems);
  OMNI_CHECK_LAUNCH();
  hipLaunchKernelGGL(teacache_decide_kernel, dim3(1), dim3(256), 0, s, *tc, n_items, rows_per_item, n_img_rows, n_txt_rows,
                     1.0f / ((float)rows_per_item * (float)D));
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

int omni_internal_teacache_post(const omni_teacache* tc, omni_bf16* hidden, const omni_bf16* hidden_in, int32_t n_img_rows,
                                int32_t rows_per_item, int32_t D, void* stream) {
  if (!tc || !hidden || !hidden_in) return OMNI_ERR_BAD_ARG;
  const int64_t nchunk = (int64_t)n_img_rows * D / 8;
  hipLaunchKernelGGL(teacache_post_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), hidden, hidden_in,
                     tc->prev_res, tc->skip, nchunk, rows_per_item * (D / 8));
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

extern "C" int omni_quantize_fp8_rows(const omni_bf16* x, int64_t ldx, int32_t x_k32_rows, int32_t rows, int32_t K,
                                      uint8_t* y8, int32_t y_rows, float* scale, omni_stream stream) {
  if (!x || !y8 || !scale || rows <= 0 || K <= 0 || y_rows < rows || x_k32_rows < 0 || (x_k32_rows && x_k32_rows < rows))
    return OMNI_ERR_BAD_ARG;
  if (K % 64 || K > 16384) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(x) || (reinterpret_cast<uintptr_t>(y8) & 7) || (!x_k32_rows && (ldx % 8))) return OMNI_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((rows + 3) / 4), block(256);
  const int nch = (K / 8 + 63) / 64;
#define OMNI_Q8(N)                                                                                                    \
  hipLaunchKernelGGL(quantize_fp8_rows_kernel<N>, grid, block, 0, s, x, ldx, x_k32_rows, rows, K, y8, y_rows, scale)
  if (nch <= 1) OMNI_Q8(1);
  else if (nch <= 2) OMNI_Q8(2);
  else if (nch <= 6) OMNI_Q8(6);
  else if (nch <= 8) OMNI_Q8(8);
  else if (nch <= 24) OMNI_Q8(24);
  else OMNI_Q8(32);
#undef OMNI_Q8
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

extern "C" int omni_adaln_modulate_ex(const omni_bf16* x, int64_t ldx, omni_bf16* y, int64_t ldy, int32_t rows,
                                      int32_t D, const omni_bf16* scale, const omni_bf16* shift,
                                      int64_t mod_item_stride, const int32_t* row_item_map, int32_t rows_per_item,
                                      float eps, int32_t y_k32_rows, omni_stream stream) {
  if (!x || !y || !scale || !shift || rows <= 0 || D <= 0) return OMNI_ERR_BAD_ARG;
  if (!row_item_map && rows_per_item <= 0) return OMNI_ERR_BAD_ARG;
  if (y_k32_rows < 0 || (y_k32_rows > 0 && y_k32_rows < rows)) return OMNI_ERR_BAD_ARG;
  if (D % 8 || D > 8192 || (y_k32_rows && D % 32)) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(x) || !omni_aligned16(y) || !omni_aligned16(scale) || !omni_aligned16(shift) || (ldx % 8) ||
      (!y_k32_rows && (ldy % 8)) || (mod_item_stride % 8))
    return OMNI_ERR_ALIGN;
  return launch_rownorm<0>(x, ldx, y, ldy, rows, D, scale, shift, mod_item_stride, row_item_map, rows_per_item, eps,
                           static_cast<hipStream_t>(stream), y_k32_rows);
}

extern "C" int omni_adaln_modulate_fp8(const omni_bf16* x, int64_t ldx, int32_t rows, int32_t D, const omni_bf16* scale,
                                       const omni_bf16* shift, int64_t mod_item_stride, const int32_t* row_item_map,
                                       int32_t rows_per_item, float eps, omni_bf16* y, int32_t y_k32_rows, uint8_t* y8,
                                       int32_t y8_rows, float* y8_scale, omni_stream stream) {
  if (!x || !y8 || !y8_scale || !scale || !shift || rows <= 0 || D <= 0 || y8_rows < rows) return OMNI_ERR_BAD_ARG;
  if (!row_item_map && rows_per_item <= 0) return OMNI_ERR_BAD_ARG;
  if (y && (y_k32_rows <= 0 || y_k32_rows < rows)) return OMNI_ERR_BAD_ARG;       // the optional bf16 copy is K32-blocked
  if (D % 64 || D > 8192) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(x) || (y && !omni_aligned16(y)) || (reinterpret_cast<uintptr_t>(y8) & 7) || !omni_aligned16(scale) ||
      !omni_aligned16(shift) || (ldx % 8) || (mod_item_stride % 8))
    return OMNI_ERR_ALIGN;
  return launch_rownorm<0>(x, ldx, y, D, rows, D, scale, shift, mod_item_stride, row_item_map, rows_per_item, eps,
                           static_cast<hipStream_t>(stream), y ? y_k32_rows : 0, y8, y8_rows, y8_scale);
}

extern "C" int omni_adaln_modulate(const omni_bf16* x, int64_t ldx, omni_bf16* y, int64_t ldy, int32_t rows,
                                   int32_t D, const omni_bf16* scale, const omni_bf16* shift,
                                   int64_t mod_item_stride, const int32_t* row_item_map, int32_t rows_per_item,
                                   float eps, omni_stream stream) {
  return omni_adaln_modulate_ex(x, ldx, y, ldy, rows, D, scale, shift, mod_item_stride, row_item_map, rows_per_item,
                                eps, 0, stream);
}

// AdaLN of the image stream and the text stream in ONE launch (ABI v12).  Both streams are row-major [rows, D] (ld = D) with
// the same modulation stride; per stream the bf16 result (row-major or K32-blocked) and / or the fp8 copy + per-row scales,
// exactly as omni_adaln_modulate_ex / omni_adaln_modulate_fp8 produce them (same kernel body, same bits).
extern "C" int omni_adaln_modulate_pair(const omni_adaln_stream* a, const omni_adaln_stream* b, int32_t D, int64_t mod_item_stride,
                                        float eps, omni_stream stream) {
  if (!a || !b || D <= 0) return OMNI_ERR_BAD_ARG;
  for (const omni_adaln_stream* g : {a, b}) {
    if (!g->x || (!g->y && !g->y8) || !g->scale || !g->shift || g->rows <= 0) return OMNI_ERR_BAD_ARG;
    if (!g->row_item_map && g->rows_per_item <= 0) return OMNI_ERR_BAD_ARG;
    if (g->y_k32_rows < 0 || (g->y_k32_rows > 0 && g->y_k32_rows < g->rows)) return OMNI_ERR_BAD_ARG;
    if (g->y8 && (!g->y8_scale || g->y8_rows < g->rows || (g->y && g->y_k32_rows <= 0))) return OMNI_ERR_BAD_ARG;
  }
  for (const omni_adaln_stream* g : {a, b})
    if (D % 8 || D > 8192 || (g->y_k32_rows && D % 32) || (g->y8 && D % 64)) return OMNI_ERR_UNSUPPORTED;
  for (const omni_adaln_stream* g : {a, b})
    if (!omni_aligned16(g->x) || (g->y && !omni_aligned16(g->y)) || !omni_aligned16(g->scale) || !omni_aligned16(g->shift) ||
        (g->y8 && (reinterpret_cast<uintptr_t>(g->y8) & 7)) || (mod_item_stride % 8))
      return OMNI_ERR_ALIGN;
  return launch_adaln_pair(*a, *b, D, mod_item_stride, eps, static_cast<hipStream_t>(stream));
}

// The finish of a deferred split-K GEMM (gated-residual epilogue) + the AdaLN behind it, both streams in one launch (ABI v13; see
// splitk_finish_adaln_kernel).
extern "C" int omni_splitk_finish_adaln_pair(const float* splitk_ws, int32_t nsplit, int64_t ws_rows,
                                             const omni_finish_adaln_stream* a, const omni_finish_adaln_stream* b, int32_t D,
                                             int64_t mod_item_stride, float eps, omni_stream stream) {
  if (!splitk_ws || !a || !b || D <= 0 || ws_rows <= 0) return OMNI_ERR_BAD_ARG;
  for (const omni_finish_adaln_stream* g : {a, b}) {
    if (!g->hidden || !g->gate || !g->scale || !g->shift || !g->y || g->rows <= 0) return OMNI_ERR_BAD_ARG;
    if (!g->row_item_map && g->rows_per_item <= 0) return OMNI_ERR_BAD_ARG;
    if (g->ws_row0 < 0 || (int64_t)g->ws_row0 + g->rows > ws_rows) return OMNI_ERR_BAD_ARG;
    if (g->y_k32_rows < 0 || (g->y_k32_rows > 0 && g->y_k32_rows < g->rows)) return OMNI_ERR_BAD_ARG;
  }
  if (nsplit != 2 && nsplit != 3 && nsplit != 4 && nsplit != 6 && nsplit != 8) return OMNI_ERR_UNSUPPORTED;
  if (D % 8 || D > 4096) return OMNI_ERR_UNSUPPORTED;
  for (const omni_finish_adaln_stream* g : {a, b}) {
    if (g->y_k32_rows && D % 32) return OMNI_ERR_UNSUPPORTED;
    if (!omni_aligned16(g->hidden) || !omni_aligned16(g->gate) || !omni_aligned16(g->scale) || !omni_aligned16(g->shift) ||
        !omni_aligned16(g->y) || (g->bias && !omni_aligned16(g->bias)))
      return OMNI_ERR_ALIGN;
  }
  if (!omni_aligned16(splitk_ws) || (mod_item_stride % 8)) return OMNI_ERR_ALIGN;
  return launch_finish_adaln(splitk_ws, nsplit, ws_rows, *a, *b, D, mod_item_stride, eps, static_cast<hipStream_t>(stream));
}

extern "C" int omni_rmsnorm(const omni_bf16* x, int64_t ldx, omni_bf16* y, int64_t ldy, int32_t rows, int32_t D,
                            const omni_bf16* weight, float eps, omni_stream stream) {
  if (!x || !y || !weight || rows <= 0 || D <= 0) return OMNI_ERR_BAD_ARG;
  if (D % 8 || D > 8192) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(x) || !omni_aligned16(y) || !omni_aligned16(weight) || (ldx % 8) || (ldy % 8))
    return OMNI_ERR_ALIGN;
  return launch_rownorm<1>(x, ldx, y, ldy, rows, D, weight, nullptr, 0, nullptr, 1, eps,
                           static_cast<hipStream_t>(stream));
}

extern "C" int omni_qk_norm_rope(omni_bf16* x, int64_t ldx, int32_t rows, int32_t num_heads, const omni_bf16* w_img,
                                 const omni_bf16* w_txt, const omni_bf16* cos_tab, const omni_bf16* sin_tab,
                                 const int32_t* row_pos, int32_t txt_pos_end, float eps, omni_stream stream) {
  if (!x || !w_img || !w_txt || !cos_tab || !sin_tab || !row_pos || rows <= 0 || num_heads <= 0)
    return OMNI_ERR_BAD_ARG;
  if (!omni_aligned16(x) || !omni_aligned16(w_img) || !omni_aligned16(w_txt) || (ldx % 8) ||
      (reinterpret_cast<uintptr_t>(cos_tab) & 7) || (reinterpret_cast<uintptr_t>(sin_tab) & 7))
    return OMNI_ERR_ALIGN;
  const int64_t units = (int64_t)rows * num_heads;
  hipLaunchKernelGGL(qk_norm_rope_kernel, dim3((unsigned)((units + 15) / 16)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, ldx, rows, num_heads, w_img, w_txt, cos_tab, sin_tab,
                     row_pos, txt_pos_end, eps);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

extern "C" int omni_rope_interleaved(const omni_bf16* x, omni_bf16* y, int32_t B, int32_t S, int32_t H, int32_t dh,
                                     const omni_bf16* cos_tab, const omni_bf16* sin_tab, omni_stream stream) {
  if (!x || !y || !cos_tab || !sin_tab || B <= 0 || S <= 0 || H <= 0 || dh <= 0) return OMNI_ERR_BAD_ARG;
  if (dh % 16) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(x) || !omni_aligned16(y) || (reinterpret_cast<uintptr_t>(cos_tab) & 7) ||
      (reinterpret_cast<uintptr_t>(sin_tab) & 7))
    return OMNI_ERR_ALIGN;
  const int64_t chunks = (int64_t)B * S * H * (dh / 8);
  hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, chunks, S, H, dh, cos_tab, sin_tab);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

extern "C" int omni_linear_smallbatch(const omni_bf16* x, int64_t ldx, int32_t B, const omni_bf16* W,
                                      const omni_bf16* bias, int64_t N, int32_t K, omni_bf16* y, int64_t ldy,
                                      int32_t act_in, int32_t act_out, omni_stream stream) {
  if (!x || !W || !y || B <= 0 || N <= 0 || K <= 0) return OMNI_ERR_BAD_ARG;
  if (B > 8 || K % 8 || K > 4096) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(W)) return OMNI_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // 2 weight rows per wave-iteration, 4 waves per block.  Grid: as many blocks as fit on the chip AT ONCE (the staged activations
  // take B * K * 4 bytes of LDS per block: 8 blocks per CU at B = 1, 3 at B = 4, 1 at B = 8), grid-striding over the rows — every
  // block stages its activations once, so more blocks than resident slots only repeat the staging (round 6: the 4-row table pass
  // ran 2048 blocks of 8 weight rows each, staging 48 KB to stream 48 KB: 1.9 TB/s)
  const int cus = omni_num_cus();
  const size_t lds_need = (size_t)B * K * sizeof(float);
  int per_cu = (int)((160 * 1024) / (lds_need + 512));
  per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
  int64_t blocks = (N + 7) / 8;
  if (blocks > (int64_t)cus * per_cu) blocks = (int64_t)cus * per_cu;
  const dim3 grid((unsigned)blocks), block(256);
  const int pieces = (K / 8 + 63) / 64;            // 16-byte pieces per lane and row: 6 at K = 3072
#define OMNI_LSB_C(NB, MC)                                                                                        \
  do {                                                                                                            \
    const size_t lds = (size_t)NB * K * sizeof(float);                                                            \
    if (lds > 65536) {                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_smallbatch_kernel<NB, MC>),                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                \
        return OMNI_ERR_LAUNCH;                                                                                   \
    }                                                                                                             \
    hipLaunchKernelGGL((linear_smallbatch_kernel<NB, MC>), grid, block, lds, s, x, ldx, W, bias, N, K, y, ldy,    \
                       act_in, act_out);                                                                          \
  } while (0)
#define OMNI_LSB(NB)                                                                                              \
  do {                                                                                                            \
    if (pieces <= 1) OMNI_LSB_C(NB, 1);                                                                           \
    else if (pieces <= 2) OMNI_LSB_C(NB, 2);                                                                      \
    else if (pieces <= 4) OMNI_LSB_C(NB, 4);                                                                      \
    else if (pieces <= 6) OMNI_LSB_C(NB, 6);                                                                      \
    else OMNI_LSB_C(NB, 8);                                                                                       \
  } while (0)
  if (B == 1) OMNI_LSB(1);
  else if (B == 2) OMNI_LSB(2);
  else if (B <= 4) {
    if (B == 3) OMNI_LSB(3); else OMNI_LSB(4);
  } else if (B <= 6) {
    if (B == 5) OMNI_LSB(5); else OMNI_LSB(6);
  } else {
    if (B == 7) OMNI_LSB(7); else OMNI_LSB(8);
  }
#undef OMNI_LSB
#undef OMNI_LSB_C
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

extern "C" int omni_timestep_sinusoid(const float* t, int32_t B, int32_t dim, float scale, omni_bf16* out,
                                      omni_stream stream) {
  if (!t || !out || B <= 0 || dim <= 0 || dim % 2) return OMNI_ERR_BAD_ARG;
  const int n = B * (dim / 2);
  hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3((n + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream),
                     t, B, dim, scale, out);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

extern "C" int omni_cfg_euler_step_ex(const omni_bf16* pos, const omni_bf16* neg, omni_bf16* latents, int32_t rows,
                                      int32_t C, float true_cfg_scale, const float* dt, int32_t dt_rows_per_item,
                                      int32_t normalize, omni_stream stream) {
  if (!pos || !latents || !dt || rows <= 0) return OMNI_ERR_BAD_ARG;
  if (C != 64) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(pos) || !omni_aligned16(latents) || (neg && !omni_aligned16(neg))) return OMNI_ERR_ALIGN;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3((rows + 31) / 32), dim3(256), 0, static_cast<hipStream_t>(stream), pos,
                     neg, latents, rows, true_cfg_scale, dt, dt_rows_per_item, normalize ? 1 : 0);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

extern "C" int omni_cfg_euler_step(const omni_bf16* pos, const omni_bf16* neg, omni_bf16* latents, int32_t rows,
                                   int32_t C, float true_cfg_scale, const float* dt, int32_t dt_rows_per_item,
                                   omni_stream stream) {
  return omni_cfg_euler_step_ex(pos, neg, latents, rows, C, true_cfg_scale, dt, dt_rows_per_item, 1, stream);
}

int omni_internal_silu_bf16(omni_bf16* dst, const omni_bf16* src, int64_t n, void* stream) {
  if (!dst || !src || n <= 0) return OMNI_ERR_BAD_ARG;
  hipLaunchKernelGGL(silu_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dst, src, n);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

int omni_internal_add_bf16(omni_bf16* dst, const omni_bf16* src, int64_t n, void* stream) {
  if (!dst || !src || n <= 0) return OMNI_ERR_BAD_ARG;
  hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dst, src, n);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}
