// Flash attention forward for gfx950: non-causal, no mask, head_dim 128, bf16 in/out, fp32 softmax.
//
// One workgroup = 4 waves = 128 query rows of one (item, head); each wave owns 32 queries.  K/V tiles of
// 64 keys are staged through LDS (2 stages x (16 KiB K + 16 KiB V) = 64 KiB -> 2 workgroups / CU).
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
//   Loads are register-staged and split (issue global loads before the MFMAs, ds_write after them).
//
// Layout contract: q/k/v/out are [total_rows, H*128] (the reference's [B,S,H,dh] flattened), item b owns rows
// [cu_seqlens[b], cu_seqlens[b+1]).  Roofline: MFMA-bound, 4*S^2*128 flop per (item, head).
#include "common.h"

namespace {

constexpr int DH = 128;
constexpr int NWAVES = 4;
constexpr int QBLK = 32 * NWAVES;  // 128 queries / workgroup
constexpr int KVBLK = 64;
constexpr int K_TILE_BYTES = KVBLK * DH * 2;  // 16 KiB
constexpr int STAGE_BYTES = 2 * K_TILE_BYTES; // K + V
constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // 64 KiB

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;

OMNI_DEVINL bf16x8_t tr_read_pair(uint32_t lds_addr_a, uint32_t lds_addr_b) {
  // two hardware-transposed 4x(16 lanes) reads -> the 8 k-elements of one MFMA A fragment
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)lds_addr_a);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)lds_addr_b);
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(NWAVES * 64, 2) void flash_attn_fwd_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    uint16_t* __restrict__ out, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
    const int32_t* __restrict__ cu_seqlens, int n_heads_total, int H, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  // heads fastest -> block b runs on XCD b%8, so every q-block of a head lands on the same XCD's L2
  const int hb = blockIdx.x % n_heads_total;
  const int qb = blockIdx.x / n_heads_total;
  const int b = hb / H, h = hb - b * H;
  const int seq_start = cu_seqlens[b];
  const int seq_len = cu_seqlens[b + 1] - seq_start;
  if (qb * QBLK >= seq_len) return;

  const uint16_t* kbase = k + (int64_t)seq_start * ldk + h * DH;
  const uint16_t* vbase = v + (int64_t)seq_start * ldv + h * DH;

  // ---- Q fragments (B operand): lane holds q = l31, d = ks*16 + hi*8 .. +8 ---------------------
  bf16x8_t qf[8];
  {
    const int qrow = min(qb * QBLK + wave * 32 + l31, seq_len - 1);
    const uint16_t* qp = q + (int64_t)(seq_start + qrow) * ldq + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
  }

  // ---- staging maps: thread t moves 16-B chunks id = t + 256*i; key = id>>4, c = id&15 ---------
  u32x4_t kreg[4], vreg[4];
  uint32_t k_wr_off[4], v_wr_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    const int key = id >> 4, c = id & 15;
    k_wr_off[i] = key * 256 + ((c ^ (key & 15)) << 4);
    v_wr_off[i] = K_TILE_BYTES + (c >> 2) * 4096 + (key >> 2) * 256 + (key & 3) * 64 + (c & 3) * 16;
  }
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + 256 * i;
      const int key = min(kv0 + (id >> 4), seq_len - 1), c = id & 15;
      kreg[i] = *reinterpret_cast<const u32x4_t*>(kbase + (int64_t)key * ldk + c * 8);
      vreg[i] = *reinterpret_cast<const u32x4_t*>(vbase + (int64_t)key * ldv + c * 8);
    }
  };
  auto store_tile = [&](int stage) {
    char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4_t*>(sb + k_wr_off[i]) = kreg[i];
      *reinterpret_cast<u32x4_t*>(sb + v_wr_off[i]) = vreg[i];
    }
  };

  // ---- fragment read addresses ------------------------------------------------------------------
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // K: row key = j*32 + l31, chunk (ks*2+hi) ^ (key&15)      (j*32 keeps key&15 = l31&15)
  const uint32_t k_row_off = l31 * 256;
  const uint32_t k_swz = l31 & 15;
  // V (tr read): g = lane>>4, m = lane&15:  base = (m>>2)*64 + (g&1)*32 + (m&3)*8 + hi*256
  const uint32_t v_lane_off = K_TILE_BYTES + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8 + hi * 256;

  f32x16_t o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;

  const int ntiles = (seq_len + KVBLK - 1) / KVBLK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const int kv0 = t * KVBLK;
    if (t + 1 < ntiles) load_tile(kv0 + KVBLK);  // in flight under this tile's MFMAs
    const char* sb = smem + cur * STAGE_BYTES;

    // ---- Sᵀ = K Qᵀ : two 32-key sub-blocks ----------------------------------------------------
    f32x16_t s[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[j][i] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const uint32_t ch = (ks * 2 + hi) ^ k_swz;
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sb + j * 32 * 256 + k_row_off + (ch << 4));
        s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[j], 0, 0, 0);
      }
    }
    // ---- mask the ragged tail (last tile only; wave-uniform branch) ---------------------------
    if (kv0 + KVBLK > seq_len) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= seq_len) s[j][r] = -INFINITY;
        }
    }
    // ---- online softmax (lane-local; one cross-half exchange for the max) ---------------------
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
    const float mneg = -m_new * scale_log2e;
    m_run = m_new;
    float psum = 0.0f;
    bf16x8_t pf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][ss * 8 + e], scale_log2e, mneg));
          psum += p;
          pf[j][ss][e] = (__bf16)p;
        }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[d][i] *= alpha;

    // ---- Oᵀ += Vᵀ Pᵀ ----------------------------------------------------------------------------
    const uint32_t vb = lds0 + cur * STAGE_BYTES + v_lane_off;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const uint32_t a0 = vb + d * 4096 + (j * 8 + ss * 4) * 256;
          const bf16x8_t vf = tr_read_pair(a0, a0 + 512);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[j][ss], o[d], 0, 0, 0);
        }

    if (t + 1 < ntiles) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: O[q][d] = Oᵀ / l ; lane holds q = l31, d = dblk*32 + 8*qd + 4*hi + {0..3} -------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = qb * QBLK + wave * 32 + l31;
  if (qrow < seq_len) {
    uint16_t* op = out + (int64_t)(seq_start + qrow) * ldo + h * DH + hi * 4;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        u32x2_t w;
        w[0] = pack_bf16x2(o[d][qd * 4 + 0] * inv, o[d][qd * 4 + 1] * inv);
        w[1] = pack_bf16x2(o[d][qd * 4 + 2] * inv, o[d][qd * 4 + 3] * inv);
        *reinterpret_cast<u32x2_t*>(op + d * 32 + qd * 8) = w;
      }
  }
}

}  // namespace

extern "C" int omni_flash_attn_fwd(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out,
                                   int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* cu_seqlens,
                                   int32_t B, int32_t H, int32_t head_dim, int32_t max_seqlen, float softmax_scale,
                                   omni_stream stream) {
  if (!q || !k || !v || !out || !cu_seqlens || B <= 0 || H <= 0 || max_seqlen <= 0) return OMNI_ERR_BAD_ARG;
  if (head_dim != DH) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(q) || !omni_aligned16(k) || !omni_aligned16(v) || (reinterpret_cast<uintptr_t>(out) & 7) ||
      (ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 4))
    return OMNI_ERR_ALIGN;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_fwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return OMNI_ERR_LAUNCH;
    attr_set = true;
  }
  const int qblocks = (max_seqlen + QBLK - 1) / QBLK;
  const int nh = B * H;
  hipLaunchKernelGGL(flash_attn_fwd_kernel, dim3(nh * qblocks), dim3(NWAVES * 64), LDS_BYTES,
                     static_cast<hipStream_t>(stream), q, k, v, out, ldq, ldk, ldv, ldo, cu_seqlens, nh, H,
                     softmax_scale * 1.4426950408889634f);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}
