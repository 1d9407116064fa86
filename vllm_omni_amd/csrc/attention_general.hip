// General flash attention forward for gfx950: everything `F.scaled_dot_product_attention` accepts at the reference's
// attention plug-in point (vllm_omni/diffusion/attention/backends/sdpa.py:46-66) that the two tuned kernels of the Qwen-Image
// hot path (attention.hip, attention_w64.hip: self-attention, head size 128, no mask) do not take:
//   * separate query / key sequences (cross-attention: wan2_2_transformer.py:243,340 attends video tokens to text tokens),
//   * head size 64 as well as 128 (sd3_transformer.py:108),
//   * attention masks: boolean (True = attend) or additive (bf16 / fp32), any layout broadcastable to [B, H, S_q, S_k]
//     (element strides, 0 = broadcast) — key-padding masks, block masks, dense biases,
//   * is_causal (top-left aligned like torch: key index <= query index),
//   * grouped K / V heads (num_kv_heads < num_heads).
// Same arithmetic scheme as attention.hip, with hipcc scheduling the loop (no hand-placed waits: this is the general path, not
// the roofline kernel):
//   one workgroup = 4 waves x 32 queries of one (item, head); K / V tiles of 64 keys staged global -> registers -> LDS, the
//   next tile's global loads in flight while the current tile is consumed;
//   S^T = K Q^T with SWAPPED operands (v_mfma_f32_32x32x16_bf16: A = K fragment, B = Q fragment kept in registers), so a lane
//   owns ONE query and 16 of a 32-key sub-block's scores: scale, mask, running max / sum are lane-local, the two half-waves
//   exchange one value per tile (v_permlane32_swap);
//   O^T += V^T P^T: B = P^T is the exp'ed S^T accumulator packed to bf16 (its C-layout key order is adopted as the MFMA k
//   order), A = V^T fragments by ds_read_b64_tr_b16 from a row-major V tile in the blocked image [d/32][key/4][4 keys][32 d].
//   fp32 softmax in the exp2 domain; scores are scaled in fp32 (q is NOT pre-rounded with the scale).
// Rows whose every key is masked produce zeros (torch's math path yields NaN there, its fused paths zeros).
// Roofline: MFMA-bound, 4 * S_q * S_k * dh flop per (item, head); measured numbers in DESIGN.md (kernel table).
#include "common.h"

namespace {

constexpr int GQ_WAVES = 4;
constexpr int GQBLK = 32 * GQ_WAVES;   // queries per workgroup
constexpr int GKV = 64;                // keys per tile

typedef __attribute__((ext_vector_type(4))) short g_s16x4_t;
typedef __attribute__((address_space(3))) g_s16x4_t g_lds_s16x4;

OMNI_DEVINL bf16x8_t g_tr_read_pair(uint32_t a, uint32_t b) {
  g_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g_lds_s16x4*)(uintptr_t)a);
  g_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g_lds_s16x4*)(uintptr_t)b);
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}
OMNI_DEVINL float g_xhalf_max(float x) {
  uint32_t a = __builtin_bit_cast(uint32_t, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
OMNI_DEVINL float g_xhalf_sum(float x) {
  uint32_t a = __builtin_bit_cast(uint32_t, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

template <int DH>
__global__ __launch_bounds__(GQ_WAVES * 64) void flash_attn_general_kernel(const omni_attn_params P, int qblocks) {
  constexpr int NCH = DH / 8;              // 16-byte chunks per K / V row
  constexpr int ROWB = DH * 2;             // bytes per K row in LDS
  constexpr int NKS = DH / 16;             // MFMA k-steps of the QK^T product
  constexpr int NDB = DH / 32;             // 32-row blocks of O^T
  constexpr int K_BYTES = GKV * ROWB;
  constexpr int NLD = GKV * NCH / (GQ_WAVES * 64);   // 16-byte pieces per thread per operand per tile (4 at dh 128, 2 at dh 64)
  __shared__ __attribute__((aligned(16))) char smem[2 * K_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;

  const int hb = blockIdx.x / qblocks, qb = blockIdx.x - hb * qblocks;
  const int b = hb / P.H, h = hb - b * P.H;
  const int hk = h / (P.H / P.H_kv);
  const int q_start = P.cu_seqlens_q[b], q_len = P.cu_seqlens_q[b + 1] - q_start;
  const int k_start = P.cu_seqlens_k[b], k_len = P.cu_seqlens_k[b + 1] - k_start;
  const int q0 = qb * GQBLK;
  if (q0 >= q_len) return;
  const int qi = q0 + wave * 32 + l31;                       // this lane's query (within the item); rows past q_len are clamped
  const int qrow = min(qi, q_len - 1);

  // Q fragments: B operand of S^T = K Q^T — lane (query l31, half hi) holds k = 16 ks + 8 hi .. + 8
  bf16x8_t qf[NKS];
  {
    const uint16_t* qp = P.q + (int64_t)(q_start + qrow) * P.ldq + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
  }
  f32x16_t o[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  const float sl2 = P.softmax_scale * 1.4426950408889634f;

  // keys this workgroup has to visit: all of them, or (causal) those up to its last query
  int k_end = k_len;
  if (P.causal) k_end = min(k_len, q0 + GQBLK);
  const int ntiles = (k_end + GKV - 1) / GKV;

  const uint16_t* kbase = P.k + (int64_t)k_start * P.ldk + hk * DH;
  const uint16_t* vbase = P.v + (int64_t)k_start * P.ldv + hk * DH;
  u32x4_t kreg[NLD], vreg[NLD];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * (GQ_WAVES * 64);
      const int key = min(t * GKV + c / NCH, k_len - 1), ch = c % NCH;      // keys past the end: re-read the last one (masked below)
      kreg[i] = *reinterpret_cast<const u32x4_t*>(kbase + (int64_t)key * P.ldk + ch * 8);
      vreg[i] = *reinterpret_cast<const u32x4_t*>(vbase + (int64_t)key * P.ldv + ch * 8);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * (GQ_WAVES * 64);
      const int key = c / NCH, ch = c % NCH;
      // K: row-major [64][DH], 16-B chunk index XOR (key & (NCH-1)): conflict-free ds_read_b128 of 32 keys at one logical chunk
      *reinterpret_cast<u32x4_t*>(smem + key * ROWB + ((ch ^ (key & (NCH - 1))) << 4)) = kreg[i];
      // V: [d/32][key/4][4 keys][32 d]: a tr-read's 16-lane group covers 4 keys x 16 d
      *reinterpret_cast<u32x4_t*>(smem + K_BYTES + (ch >> 2) * 4096 + (key >> 2) * 256 + (key & 3) * 64 + (ch & 3) * 16) = vreg[i];
    }
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t k_addr[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) k_addr[ks] = lds0 + l31 * ROWB + ((((uint32_t)(ks * 2 + hi)) ^ (l31 & (NCH - 1))) << 4);
  const uint32_t v_addr = lds0 + K_BYTES + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8 + hi * 256;

  const char* mrow = nullptr;              // this lane's mask row (element (b, h, qi, 0))
  const int msz = P.mask_type == 1 ? 1 : (P.mask_type == 2 ? 2 : 4);
  if (P.mask_type) mrow = reinterpret_cast<const char*>(P.mask) +
                          ((int64_t)b * P.mask_stride_b + (int64_t)h * P.mask_stride_h + (int64_t)qrow * P.mask_stride_q) * msz;

  if (ntiles > 0) {
    load_tile(0);
    store_tile();
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) load_tile(t + 1);                    // in flight while tile t is consumed
    // ---- S^T = K Q^T: two 32-key sub-blocks
    typedef __attribute__((address_space(3))) const bf16x8_t lds_bf16x8;
    f32x16_t s[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[j][i] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<lds_bf16x8*>((uintptr_t)(k_addr[ks] + j * 32 * ROWB));
        s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[j], 0, 0, 0);
      }
    }
    // ---- scale, mask, online softmax.  register r of sub-block j <-> key t*64 + 32 j + (r & 3) + 8 (r >> 2) + 4 hi
    const int kv0 = t * GKV;
    // which tiles need per-element work: the ragged last tile; with `causal` only the tiles that reach past this WAVE's smallest
    // query index (everything before the diagonal is attended as it is); with a mask every tile
    const bool tail = kv0 + GKV > k_len;
    const bool diag = P.causal && kv0 + GKV - 1 > q0 + wave * 32;
    float mx = -INFINITY;
    if (!(tail || diag || P.mask_type)) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[j][r] *= sl2;
          mx = fmaxf(mx, s[j][r]);
        }
    } else {
      // mask values of a lane's four CONSECUTIVE keys (r & 3) come in one load when the mask is contiguous along the keys and
      // the address allows it (the usual cases: a [B, 1, 1, S_k] padding mask, a dense [.., S_q, S_k] bias); else one by one
      const bool vec = P.mask_type && P.mask_stride_k == 1;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int kb = kv0 + j * 32 + 8 * rq + 4 * hi;                 // first of this lane's four consecutive keys
          float add[4] = {0.f, 0.f, 0.f, 0.f};
          bool ok[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) ok[e] = kb + e < k_len && (!P.causal || kb + e <= qi);
          if (P.mask_type && kb < k_len) {
            const char* mp = mrow + (int64_t)kb * P.mask_stride_k * msz;
            const bool whole = vec && kb + 3 < k_len && (reinterpret_cast<uintptr_t>(mp) & (4 * msz - 1)) == 0;
            if (P.mask_type == 1) {
              uint32_t w = 0;
              if (whole) w = *reinterpret_cast<const uint32_t*>(mp);
              else
                for (int e = 0; e < 4; ++e)
                  if (kb + e < k_len) w |= (uint32_t)(*reinterpret_cast<const uint8_t*>(mp + (int64_t)e * P.mask_stride_k) != 0) << (8 * e);
#pragma unroll
              for (int e = 0; e < 4; ++e) ok[e] = ok[e] && ((w >> (8 * e)) & 0xffu) != 0;
            } else if (P.mask_type == 2) {
              if (whole) {
                const u32x2_t w = *reinterpret_cast<const u32x2_t*>(mp);
                add[0] = bf16_lo(w[0]); add[1] = bf16_hi(w[0]); add[2] = bf16_lo(w[1]); add[3] = bf16_hi(w[1]);
              } else {
                for (int e = 0; e < 4; ++e)
                  if (kb + e < k_len) add[e] = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(mp + (int64_t)e * P.mask_stride_k * 2));
              }
            } else {
              if (whole) {
                const f32x4_t w = *reinterpret_cast<const f32x4_t*>(mp);
                add[0] = w[0]; add[1] = w[1]; add[2] = w[2]; add[3] = w[3];
              } else {
                for (int e = 0; e < 4; ++e)
                  if (kb + e < k_len) add[e] = *reinterpret_cast<const float*>(mp + (int64_t)e * P.mask_stride_k * 4);
              }
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = ok[e] ? s[j][rq * 4 + e] * sl2 + add[e] * 1.4426950408889634f : -INFINITY;
            s[j][rq * 4 + e] = x;
            mx = fmaxf(mx, x);
          }
        }
    }
    mx = g_xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float m_use = m_new == -INFINITY ? 0.0f : m_new;   // nothing attendable so far: every p below is exp2(-inf) = 0
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    m_run = m_new;
    float psum = 0.0f;
    uint32_t pk[2][2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(s[j][r] - m_use), p1 = __builtin_amdgcn_exp2f(s[j][r + 1] - m_use);
        psum += p0 + p1;
        pk[j][r >> 3][(r >> 1) & 3] = pack_bf16x2(p0, p1);
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
    // ---- O^T += V^T P^T: four 16-key steps (sub-block j, half hf) x NDB 32-row blocks of d
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const u32x4_t p4 = {pk[j][hf][0], pk[j][hf][1], pk[j][hf][2], pk[j][hf][3]};
        const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, p4);
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
          const uint32_t va = v_addr + d * 4096 + (j * 8 + hf * 4) * 256;
          const bf16x8_t vf = g_tr_read_pair(va, va + 512);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
        }
      }
    __syncthreads();                                         // every wave is done with tile t
    if (t + 1 < ntiles) {
      store_tile();
      __syncthreads();
    }
  }

  // ---- epilogue: lane (query l31, half hi), register r of block d <-> channel 32 d + (r & 3) + 8 (r >> 2) + 4 hi
  const float l = g_xhalf_sum(l_run);
  const float inv = l > 0.0f ? 1.0f / l : 0.0f;
  if (qi < q_len) {
    uint16_t* op = P.out + (int64_t)(q_start + qi) * P.ldo + h * DH + hi * 4;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        u32x2_t w;
        w[0] = pack_bf16x2(o[d][qd * 4 + 0] * inv, o[d][qd * 4 + 1] * inv);
        w[1] = pack_bf16x2(o[d][qd * 4 + 2] * inv, o[d][qd * 4 + 3] * inv);
        *reinterpret_cast<u32x2_t*>(op + d * 32 + qd * 8) = w;
      }
  }
}

template <int DH>
int launch_general(const omni_attn_params* p, hipStream_t s) {
  const int qblocks = (p->max_seqlen_q + GQBLK - 1) / GQBLK;
  const long grid = (long)p->B * p->H * qblocks;
  if (grid <= 0 || grid > 0x7fffffffL) return OMNI_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((flash_attn_general_kernel<DH>), dim3((unsigned)grid), dim3(GQ_WAVES * 64), 0, s, *p, qblocks);
  OMNI_CHECK_LAUNCH();
  return OMNI_OK;
}

}  // namespace

extern "C" int omni_flash_attn_general(const omni_attn_params* p, omni_stream stream) {
  if (!p || !p->q || !p->k || !p->v || !p->out || !p->cu_seqlens_q || !p->cu_seqlens_k) return OMNI_ERR_BAD_ARG;
  if (p->B <= 0 || p->H <= 0 || p->H_kv <= 0 || p->max_seqlen_q <= 0 || p->max_seqlen_k < 0) return OMNI_ERR_BAD_ARG;
  if (p->H % p->H_kv != 0) return OMNI_ERR_BAD_ARG;
  if (p->causal != 0 && p->causal != 1) return OMNI_ERR_BAD_ARG;
  if (p->mask_type < 0 || p->mask_type > 3 || (p->mask_type != 0) != (p->mask != nullptr)) return OMNI_ERR_BAD_ARG;
  if (p->head_dim != 64 && p->head_dim != 128) return OMNI_ERR_UNSUPPORTED;
  if (!omni_aligned16(p->q) || !omni_aligned16(p->k) || !omni_aligned16(p->v) || (reinterpret_cast<uintptr_t>(p->out) & 7) ||
      (p->ldq % 8) || (p->ldk % 8) || (p->ldv % 8) || (p->ldo % 4))
    return OMNI_ERR_ALIGN;
  if (p->mask_type == 2 && (reinterpret_cast<uintptr_t>(p->mask) & 1)) return OMNI_ERR_ALIGN;
  if (p->mask_type == 3 && (reinterpret_cast<uintptr_t>(p->mask) & 3)) return OMNI_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return p->head_dim == 128 ? launch_general<128>(p, s) : launch_general<64>(p, s);
}
