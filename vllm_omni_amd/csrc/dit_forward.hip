// Host-side runner: enqueues one full QwenImageTransformer2DModel.forward (reference
// vllm_omni/diffusion/models/qwen_image/qwen_image_transformer.py:692-802) on a HIP stream from a descriptor of
// device pointers.  No allocation, no host sync: the sequence is hipGraph-capturable.
//
// Batch layout (ragged, token-major): image rows [n_img_rows, D] and text rows [n_txt_rows, D] are separate
// residual streams; q/k/v/attention live in the JOINT order ([text_i ; image_i] per item, cu_seqlens) so the
// reference's three torch.cat (:414-416) and the split (:448-449) never materialise: the QKV GEMM scatters its
// rows into the joint buffers and the out-projection GEMM gathers them back.
#include <stdio.h>

#include <stdlib.h>

#include "common.h"

namespace {

struct Workspace {
  omni_bf16 *tproj, *th, *temb, *mod_img, *mod_txt, *emb_out;
  omni_bf16 *hidden_img, *hidden_txt, *xn, *txt_normed, *q, *k, *v, *attn, *mlp_h, *h_in;
  int32_t *img_pos, *txt_pos;   // RoPE table row of every image / text stream row (joint_pos gathered through the joint-row maps)
  float* splitk;                // fp32 partial tiles of the split-K GEMMs (small batches: whole-launch split-K; larger ones: the
  int64_t splitk_floats;        // tail split of a thin last round — include/omni_cdna4.h splitk_ws, OMNI_GEMM_KERNEL_NO_TAIL_SPLIT)
  int64_t splitk_tail_floats;   // the buffer's whole size (>= splitk_floats): what MLP-up's thin-tail split may use
  void* attn_part;              // fp32 partials of the attention's split short last q-block (attention_w64.hip)
  size_t attn_part_bytes;
  uint8_t* x8;                  // fp8 mode: the e4m3 copy of the NEXT block GEMM's input ([K/64][rows][64]; image rows, then text rows)
  float* x8_scale;              // fp8 mode: its per-row scales [n_joint_rows]
  size_t total;
};

inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

Workspace carve(void* base, const omni_dit_weights* w, int64_t Ri, int64_t Rt, int64_t nT) {
  const int64_t D = (int64_t)w->num_heads * w->head_dim;
  const int64_t Rj = Ri + Rt;
  size_t off = 0;
  char* b = static_cast<char*>(base);
  auto take = [&](int64_t elems) {
    omni_bf16* p = reinterpret_cast<omni_bf16*>(b + off);
    off += align_up((size_t)elems * sizeof(omni_bf16));
    return p;
  };
  Workspace ws;
  ws.tproj = take(nT * 256);
  ws.th = take(nT * D);
  ws.temb = take(nT * D);
  ws.mod_img = take(nT * 6 * D);
  ws.mod_txt = take(nT * 6 * D);
  ws.emb_out = take(nT * 2 * D);
  ws.hidden_img = take(Ri * D);
  ws.hidden_txt = take(Rt * D);
  ws.xn = take(Rj * D);
  ws.txt_normed = take(Rt * w->joint_dim);
  ws.q = take(Rj * D);
  ws.k = take(Rj * D);
  ws.v = take(Rj * D);
  ws.attn = take(Rj * D);
  ws.mlp_h = take(Rj * 4 * D);
  ws.h_in = take(Ri * D);           // image stream at block-stack entry (TeaCache residual / skip path)
  ws.img_pos = reinterpret_cast<int32_t*>(take(Ri * 2));
  ws.txt_pos = reinterpret_cast<int32_t*>(take(Rt * 2));
  // split-K workspace of the GEMMs whose grid would leave half of the chip idle (include/omni_cdna4.h splitk_ws): up to 8
  // splits x rows x D fp32 for a batch of at most four row tiles (63 MB for a CFG pair of 256x256 images), 2 splits up to ten
  // row tiles (a CFG pair at 512x512: 53 MB); larger batches fill the chip without it, but their LAST round of tiles may be thin
  // (1548 tiles = 6 rounds + 12 at one 2048^2 request): the tail split takes up to 64 tail tiles x 8 pieces of 256 KiB (134 MB)
  ws.splitk_floats = Rj <= 4 * 256 ? 8 * Rj * D : (Rj <= 10 * 256 ? 2 * Rj * D : (int64_t)512 * 256 * 256);
  // (the size handed to the N = 3072 GEMMs also CAPS their split factor — 2-way up to ten row tiles — so MLP-up's thin-tail split,
  // run_block / mlp_up_tail_split, gets its own figure over the same buffer: a 64-tile tail x 4 pieces of 256 KiB = 67 MB)
  ws.splitk_tail_floats = ws.splitk_floats < (int64_t)256 * 256 * 256 ? (int64_t)256 * 256 * 256 : ws.splitk_floats;
  ws.splitk = reinterpret_cast<float*>(take(ws.splitk_tail_floats * 2));
  ws.attn_part_bytes = (size_t)512 * 64 * (128 + 2) * sizeof(float);      // up to 512 (item, head, key range) partials of 64 rows
  ws.attn_part = take((int64_t)(ws.attn_part_bytes / 2));
  ws.x8 = nullptr; ws.x8_scale = nullptr;
  if (w->fp8_layers) {
    ws.x8 = reinterpret_cast<uint8_t*>(take(Rj * 2 * D));             // Rj x 4D bytes (the MLP-down input is the widest)
    ws.x8_scale = reinterpret_cast<float*>(take(Rj * 2));
  }
  ws.total = off;
  return ws;
}

#define OMNI_TRY(expr)            \
  do {                            \
    const int _st = (expr);       \
    if (_st != OMNI_OK) return _st; \
  } while (0)


// Activations that only travel from one of this library's kernels into a GEMM A operand (AdaLN output, attention output,
// GELU output) are kept K32-blocked ([K/32][rows][32], same bytes) so that the GEMM's LDS-DMA pieces fetch whole cache
// lines (include/omni_cdna4.h: omni_gemm_group.a_k32_rows).  The residual stream, q/k/v and everything the caller sees stay
// row-major.  dev knob: OMNI_DIT_ACT_BLOCKED=0.
bool dit_act_blocked() {
  static const bool v = omni_dev_env_int("OMNI_DIT_ACT_BLOCKED", 1) != 0;
  return v;
}
// q/k RMSNorm + RoPE inside the QKV GEMM's coalesced epilogue (saves two passes over q and k per layer); dev knob
// OMNI_DIT_FUSE_QKROPE=0 restores the separate omni_qk_norm_rope launches (bit-identical results).
bool dit_fuse_qkrope() {
  static const bool v = omni_dev_env_int("OMNI_DIT_FUSE_QKROPE", 1) != 0;
  return v;
}

// omni_linear_smallbatch takes at most 8 rows (one weight stream shared by 8 activations): more conditioning rows (a
// reference-shaped forward over B > 8 items with their own timesteps) run in chunks of 8, each re-streaming the weights
int linear_rows(const omni_bf16* x, int64_t ldx, int32_t B, const omni_bf16* W, const omni_bf16* bias, int64_t N, int32_t K,
                omni_bf16* y, int64_t ldy, int32_t act_in, int32_t act_out, omni_stream stream) {
  for (int32_t b0 = 0; b0 < B; b0 += 8) {
    const int32_t nb = B - b0 < 8 ? B - b0 : 8;
    const int st = omni_linear_smallbatch(x + (int64_t)b0 * ldx, ldx, nb, W, bias, N, K, y + (int64_t)b0 * ldy, ldy, act_in,
                                          act_out, stream);
    if (st != OMNI_OK) return st;
  }
  return OMNI_OK;
}

struct BlockPred { const int32_t *tile_img, *tile_txt, *item; };

// Split-K finish folded into the AdaLN behind it (ABI v13, omni_splitk_finish_adaln_pair): when the out-projection / the MLP
// down-projection of a block runs K-split (a forward over one or two small images), its finish kernel is not launched — the
// AdaLN that follows (norm2 of the same block / norm1 of the NEXT block) sums the partials, applies bias + gate + residual,
// writes the residual stream and normalises the row in the same pass.  Same bits as the three-kernel sequence.  -DOMNI_DEV
// builds: OMNI_DIT_FUSE_FINISH = 0 off, 1 out-projection only, 2 MLP-down only, 3 both (the product: 3, no switch).
int dit_fuse_finish() {
  static const int v = omni_dev_env_int("OMNI_DIT_FUSE_FINISH", 3);
  return v;
}
// MLP-up (N = 4 D: the launch with the most tiles) gets the split-K workspace — i.e. the GEMM's TAIL SPLIT — only when a thin tail
// (<= a quarter round) follows ONE to THREE full rounds of the CUs.  One 384^2 CFG pair (1152 + 128 rows: 6 x 48 = 288 tiles = 1
// round + 32) spent HALF of the launch on a round that is an eighth full: 27.4 -> 26.3 ms per 60-layer forward (-4.2 %); 2 rounds + 64
// (one 576^2 request) -0.8 %, 3 + 48 (704^2) -0.5 % (profiles/r06l_ab_mlpup_tail_rounds.log).  With six or more full rounds in
// front of it the same split gains nothing inside a forward (profiles/r06b_ab_mlp_up_tail_split_rejected.log), and at <= 128 tiles
// the workspace would switch the whole launch to split-K with N = 4 D wide partials.  The tail tiles then carry the tail split's
// summation order (include/omni_cdna4.h, ABI v11).  -DOMNI_DEV builds: OMNI_DIT_MLPUP_TAIL = the largest number of rounds, 0 off.
bool mlp_up_tail_split(int32_t Ri, int32_t Rt, int64_t N) {
  static const int knob = omni_dev_env_int("OMNI_DIT_MLPUP_TAIL", 3);
  if (!knob) return false;
  const int64_t tiles = ((int64_t)(Ri + 255) / 256 + (Rt + 255) / 256) * ((N + 255) / 256);
  const int cus = omni_num_cus();
  const int64_t rounds = tiles / cus, tail = tiles % cus;
  return rounds >= 1 && rounds <= knob && tail > 0 && tail <= cus / 4;
}
// an MLP down-projection whose finish is still pending when its block returns (the next block's norm1 performs it)
struct PendingFinish {
  int nsplit = 0;
  const omni_bf16 *bias_img = nullptr, *bias_txt = nullptr, *gate_img = nullptr, *gate_txt = nullptr;
};

// One dual-stream block (reference QwenImageTransformerBlock.forward, qwen_image_transformer.py:541-605) on the residual
// streams hidden_img / hidden_txt (in place).  `after_img_norm1` (nullable) runs right after the image stream's first AdaLN:
// omni_dit_forward hooks the TeaCache decision there (the "modulated input" of extractors.py:189-194).
enum BlockPhase { BLOCK_ALL = 0, BLOCK_QKV = 1, BLOCK_POST = 2 };

// phase: BLOCK_ALL = the whole block; BLOCK_QKV = modulation, norm1, fused QKV projection (+ q/k norm + RoPE) into the joint
// q/k/v buffers of the workspace, then stop; BLOCK_POST = from the output projections on, reading the attention output from
// `attn_in` (ROW-MAJOR [n_joint_rows, D]) — the two halves a sequence-parallel caller runs around its all-to-alls
// (reference attention/parallel/ulysses.py:59-135).  BLOCK_POST recomputes the block's modulation vectors (two GEMVs).
// AdaLN of the two streams of a block: ONE launch (round 6, omni_adaln_modulate_pair).  A -DOMNI_DEV build can fall back to one
// launch per stream (OMNI_DIT_ADALN_PAIR=0) for same-box A/B runs; the product has no switch.
int adaln_streams(const omni_adaln_stream& si, const omni_adaln_stream& st, int32_t D, float eps, omni_stream stream) {
  static const int pair = omni_dev_env_int("OMNI_DIT_ADALN_PAIR", 1);
  if (pair) return omni_adaln_modulate_pair(&si, &st, D, 6 * (int64_t)D, eps, stream);
  for (const omni_adaln_stream* g : {&si, &st}) {
    if (g->y8)
      OMNI_TRY(omni_adaln_modulate_fp8(g->x, D, g->rows, D, g->scale, g->shift, 6 * (int64_t)D, g->row_item_map, g->rows_per_item, eps,
                                       g->y, g->y_k32_rows, g->y8, g->y8_rows, g->y8_scale, stream));
    else
      OMNI_TRY(omni_adaln_modulate_ex(g->x, D, g->y, D, g->rows, D, g->scale, g->shift, 6 * (int64_t)D, g->row_item_map,
                                      g->rows_per_item, eps, g->y_k32_rows, stream));
  }
  return OMNI_OK;
}

template <typename Hook>
int run_block(const omni_dit_weights* w, int l, const omni_dit_batch* b, const Workspace& ws, omni_bf16* hidden_img,
              omni_bf16* hidden_txt, const omni_bf16* temb, const BlockPred& pr, Hook&& after_img_norm1, omni_stream stream,
              BlockPhase phase = BLOCK_ALL, const omni_bf16* attn_in = nullptr, PendingFinish* pend = nullptr,
              bool may_defer_down = false) {
  const omni_dit_layer_weights& L = w->layers[l];
  const int32_t Ri = b->n_img_rows, Rt = b->n_txt_rows, nT = b->n_temb;
  const int32_t D = w->num_heads * w->head_dim;
  const float eps = 1e-6f;
  omni_bf16* xn_img = ws.xn;
  omni_bf16* xn_txt = ws.xn + (int64_t)Ri * D;
  omni_bf16* h_img = ws.mlp_h;
  omni_bf16* h_txt = ws.mlp_h + (int64_t)Ri * 4 * D;
  const float sm_scale = 1.0f / sqrtf((float)w->head_dim);
  const bool fuse_qkrope = dit_fuse_qkrope();
  const bool q_prescale = fuse_qkrope && phase == BLOCK_ALL;
  const bool blk = dit_act_blocked() && (D % 32 == 0);
  const int32_t bRi = blk ? Ri : 0, bRt = blk ? Rt : 0, bRj = blk ? Ri + Rt : 0;
  // fp8 mode (omni_dit_weights.fp8_layers): every block GEMM reads e4m3 operands; its bf16 input is quantised per token first
  const omni_dit_fp8_layer* F = w->fp8_layers ? &w->fp8_layers[l] : nullptr;
  if (F && (D % 128 != 0 || !ws.x8)) return OMNI_ERR_UNSUPPORTED;
  // ABI v9: per GEMM CLASS — a class whose two weight pointers are NULL stays bf16 (mixed recipes: the classes that feed the
  // residual stream directly are the accuracy-critical ones)
  const bool f_qkv = F && F->to_qkv_w8 && F->add_qkv_w8, f_out = F && F->to_out_w8 && F->to_add_out_w8;
  const bool f_up = F && F->img_mlp_w1_8 && F->txt_mlp_w1_8, f_down = F && F->img_mlp_w2_8 && F->txt_mlp_w2_8;
  // image rows -> x8[0 .. Ri), text rows -> x8[Ri ..): two [rows, K] operands in the K64-blocked order, scales alongside
  auto quant_streams = [&](const omni_bf16* xi, int32_t xi_k32, const omni_bf16* xt, int32_t xt_k32, int32_t K) -> int {
    OMNI_TRY(omni_quantize_fp8_rows(xi, K, xi_k32, Ri, K, ws.x8, Ri, ws.x8_scale, stream));
    OMNI_TRY(omni_quantize_fp8_rows(xt, K, xt_k32, Rt, K, ws.x8 + (int64_t)Ri * K, Rt, ws.x8_scale + Ri, stream));
    return OMNI_OK;
  };
  auto fp8_streams = [&](omni_gemm_params& p, int32_t K, const uint8_t* wi8, const float* wis, const uint8_t* wt8,
                         const float* wts) {
    p.fp8 = 1; p.w_k32_blocked = 1; p.splitk_ws = nullptr; p.splitk_ws_floats = 0;
    p.g[0].A = reinterpret_cast<const omni_bf16*>(ws.x8); p.g[0].a_k32_rows = Ri; p.g[0].a_scale = ws.x8_scale;
    p.g[1].A = reinterpret_cast<const omni_bf16*>(ws.x8 + (int64_t)Ri * K); p.g[1].a_k32_rows = Rt; p.g[1].a_scale = ws.x8_scale + Ri;
    p.g[0].W = reinterpret_cast<const omni_bf16*>(wi8); p.g[0].w_scale = wis;
    p.g[1].W = reinterpret_cast<const omni_bf16*>(wt8); p.g[1].w_scale = wts;
  };
  // modulation vectors [shift1|scale1|gate1|shift2|scale2|gate2]  (reference :552-561): two weight-streaming GEMVs (226 MB
  // of weights per layer) — or, ABI v9, rows of a table computed ONCE for all the denoising steps of a request
  // (omni_dit_modulation_table: the conditioning of every step is known before the loop starts)
  const omni_bf16 *mod_img = ws.mod_img, *mod_txt = ws.mod_txt;
  if (b->mod_table) {
    mod_img = b->mod_table + (int64_t)(2 * l + 0) * nT * 6 * D;
    mod_txt = b->mod_table + (int64_t)(2 * l + 1) * nT * 6 * D;
  } else {
    OMNI_TRY(linear_rows(temb, D, nT, L.img_mod_w, L.img_mod_b, 6 * (int64_t)D, D, ws.mod_img, 6 * D, 1, 0, stream));
    OMNI_TRY(linear_rows(temb, D, nT, L.txt_mod_w, L.txt_mod_b, 6 * (int64_t)D, D, ws.mod_txt, 6 * D, 1, 0, stream));
  }
  if (phase != BLOCK_POST) {
  // norm1 + modulate (reference :564-567).  fp8 mode: the e4m3 copy + per-token scale come out of the same pass (the bf16
  // result is still written for the image stream when TeaCache reads it)
  // Both streams in ONE launch (round 6, elementwise.hip rownorm_kernel PAIR; the TeaCache hook reads the image stream's
  // result behind it).
  const bool fused_q = f_qkv && blk;
  if (pend && pend->nsplit > 1) {
    // the previous block's MLP down-projection left its K-split partials: finish + gated residual + this block's norm1 in one pass
    const omni_finish_adaln_stream fi = {Ri, 0, pend->bias_img, hidden_img, pend->gate_img, mod_img + D, mod_img, b->img_item, 0,
                                         xn_img, bRi};
    const omni_finish_adaln_stream ft = {Rt, Ri, pend->bias_txt, hidden_txt, pend->gate_txt, mod_txt + D, mod_txt, b->txt_item, 0,
                                         xn_txt, bRt};
    OMNI_TRY(omni_splitk_finish_adaln_pair(ws.splitk, pend->nsplit, (int64_t)Ri + Rt, &fi, &ft, D, 6 * (int64_t)D, eps, stream));
    pend->nsplit = 0;
  } else if (fused_q) {
    const omni_adaln_stream si = {hidden_img, b->teacache ? xn_img : nullptr, Ri, mod_img + D, mod_img, b->img_item, 0,
                                  b->teacache ? bRi : 0, ws.x8, Ri, ws.x8_scale};
    const omni_adaln_stream st = {hidden_txt, nullptr, Rt, mod_txt + D, mod_txt, b->txt_item, 0, 0, ws.x8 + (int64_t)Ri * D,
                                  Rt, ws.x8_scale + Ri};
    OMNI_TRY(adaln_streams(si, st, D, eps, stream));
  } else {
    const omni_adaln_stream si = {hidden_img, xn_img, Ri, mod_img + D, mod_img, b->img_item, 0, bRi, nullptr, 0, nullptr};
    const omni_adaln_stream st = {hidden_txt, xn_txt, Rt, mod_txt + D, mod_txt, b->txt_item, 0, bRt, nullptr, 0, nullptr};
    OMNI_TRY(adaln_streams(si, st, D, eps, stream));
  }
  OMNI_TRY(after_img_norm1(xn_img));
  // fused QKV projections of both streams, scattered into the joint q/k/v (reference :380-394, :414-416)
  {
    omni_gemm_params p = {};
    p.ngroups = 2; p.N = 3 * D; p.K = D; p.split_n = D; p.w_k32_blocked = w->gemm_w_k32_blocked;
    p.epilogue = fuse_qkrope ? OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE : OMNI_EPI_BIAS_SPLIT3;
    if (fuse_qkrope) {
      p.g[0].qk_norm_q_w = L.norm_q_w; p.g[0].qk_norm_k_w = L.norm_k_w; p.g[0].qk_row_pos = ws.img_pos;
      p.g[1].qk_norm_q_w = L.norm_added_q_w; p.g[1].qk_norm_k_w = L.norm_added_k_w; p.g[1].qk_row_pos = ws.txt_pos;
      for (int g = 0; g < 2; ++g) {
        p.g[g].qk_rope_cos = b->rope_cos; p.g[g].qk_rope_sin = b->rope_sin; p.g[g].qk_eps = eps;
        // the whole-block path hands q to the attention kernel pre-multiplied by softmax_scale * log2(e) (one rounding, as
        // before); a sequence-parallel caller (BLOCK_QKV) gets the reference's un-scaled q back
        if (q_prescale) p.g[g].qk_q_scale = sm_scale * 1.4426950408889634f;
      }
    }
    p.g[0].a_k32_rows = bRi; p.g[1].a_k32_rows = bRt;
    p.g[0].A = xn_img; p.g[0].lda = D; p.g[0].M = Ri; p.g[0].W = L.to_qkv_w; p.g[0].bias = L.to_qkv_b;
    p.g[0].out = ws.q; p.g[0].out1 = ws.k; p.g[0].out2 = ws.v; p.g[0].ldo = D; p.g[0].out_row_map = b->img_joint_row;
    p.g[1].A = xn_txt; p.g[1].lda = D; p.g[1].M = Rt; p.g[1].W = L.add_qkv_w; p.g[1].bias = L.add_qkv_b;
    p.g[1].out = ws.q; p.g[1].out1 = ws.k; p.g[1].out2 = ws.v; p.g[1].ldo = D; p.g[1].out_row_map = b->txt_joint_row;
    p.g[0].tile_skip = pr.tile_img; p.g[1].tile_skip = pr.tile_txt;
    p.splitk_ws = ws.splitk; p.splitk_ws_floats = ws.splitk_floats;
    if (f_qkv) {
      if (!fused_q) OMNI_TRY(quant_streams(xn_img, bRi, xn_txt, bRt, D));
      fp8_streams(p, D, F->to_qkv_w8, F->to_qkv_s, F->add_qkv_w8, F->add_qkv_s);
    }
    OMNI_TRY(omni_gemm_bf16(&p, stream));
  }
  // per-head RMSNorm + RoPE on q and k (reference :397-410)
  if (!fuse_qkrope) {
    OMNI_TRY(omni_qk_norm_rope(ws.q, D, Ri + Rt, w->num_heads, L.norm_q_w, L.norm_added_q_w, b->rope_cos, b->rope_sin,
                               b->joint_pos, b->txt_pos_end, eps, stream));
    OMNI_TRY(omni_qk_norm_rope(ws.k, D, Ri + Rt, w->num_heads, L.norm_k_w, L.norm_added_k_w, b->rope_cos, b->rope_sin,
                               b->joint_pos, b->txt_pos_end, eps, stream));
  }
  if (phase == BLOCK_QKV) return OMNI_OK;
  // joint attention (reference :437-443 -> attention/backends/sdpa.py:46-66)
  OMNI_TRY(omni_internal_flash_attn(ws.q, ws.k, ws.v, ws.attn, D, D, D, D, b->cu_seqlens, b->n_items, w->num_heads,
                                    w->head_dim, b->max_seqlen, sm_scale, bRj, pr.item, q_prescale ? 1 : 0, ws.attn_part,
                                    ws.attn_part_bytes, stream));
  }  // phase != BLOCK_POST
  const omni_bf16* attn_src = phase == BLOCK_POST ? attn_in : ws.attn;
  const int32_t attn_k32 = phase == BLOCK_POST ? 0 : bRj;       // a caller-provided attention output is row-major
  // split-K finishes folded into the following AdaLN: bf16 blocks without TeaCache predicates (a skipped tile leaves no partials)
  // and a width the fused kernel takes
  const bool fuse_ok = !F && !pr.tile_img && !pr.tile_txt && D % 8 == 0 && D <= 4096;
  bool norm2_done = false;
  // output projections + gated residual (reference :448-456, :586-587)
  {
    omni_gemm_params p = {};
    p.ngroups = 2; p.N = D; p.K = D; p.epilogue = OMNI_EPI_BIAS_GATE_RES; p.w_k32_blocked = w->gemm_w_k32_blocked;
    p.g[0].a_k32_rows = attn_k32; p.g[1].a_k32_rows = attn_k32;
    p.g[0].A = attn_src; p.g[0].lda = D; p.g[0].a_row_map = b->img_joint_row; p.g[0].M = Ri;
    p.g[0].W = L.to_out_w; p.g[0].bias = L.to_out_b; p.g[0].out = hidden_img; p.g[0].ldo = D;
    p.g[0].res = hidden_img; p.g[0].ldres = D; p.g[0].gate = mod_img + 2 * D; p.g[0].gate_item_stride = 6 * D;
    p.g[0].row_item_map = b->img_item;
    p.g[1].A = attn_src; p.g[1].lda = D; p.g[1].a_row_map = b->txt_joint_row; p.g[1].M = Rt;
    p.g[1].W = L.to_add_out_w; p.g[1].bias = L.to_add_out_b; p.g[1].out = hidden_txt; p.g[1].ldo = D;
    p.g[1].res = hidden_txt; p.g[1].ldres = D; p.g[1].gate = mod_txt + 2 * D; p.g[1].gate_item_stride = 6 * D;
    p.g[1].row_item_map = b->txt_item;
    p.g[0].tile_skip = pr.tile_img; p.g[1].tile_skip = pr.tile_txt;
    p.splitk_ws = ws.splitk; p.splitk_ws_floats = ws.splitk_floats;
    if (f_out) {                              // the attention output in its joint order: ONE operand, both groups gather from it
      const int32_t Rj = Ri + Rt;
      OMNI_TRY(omni_quantize_fp8_rows(attn_src, D, attn_k32, Rj, D, ws.x8, Rj, ws.x8_scale, stream));
      p.fp8 = 1; p.w_k32_blocked = 1; p.splitk_ws = nullptr; p.splitk_ws_floats = 0;
      for (int g = 0; g < 2; ++g) {
        p.g[g].A = reinterpret_cast<const omni_bf16*>(ws.x8); p.g[g].a_k32_rows = Rj; p.g[g].a_scale = ws.x8_scale;
      }
      p.g[0].W = reinterpret_cast<const omni_bf16*>(F->to_out_w8); p.g[0].w_scale = F->to_out_s;
      p.g[1].W = reinterpret_cast<const omni_bf16*>(F->to_add_out_w8); p.g[1].w_scale = F->to_add_out_s;
    }
    // K-split launch (small batches): leave the partials to norm2's pass instead of a finish kernel
    const int ns = (fuse_ok && (dit_fuse_finish() & 1)) ? omni_gemm_splitk_factor(&p) : 1;
    if (ns > 1) {
      p.kernel_hint = OMNI_GEMM_KERNEL_SPLITK_DEFER_FINISH;
      OMNI_TRY(omni_gemm_bf16(&p, stream));
      const omni_finish_adaln_stream fi = {Ri, 0, L.to_out_b, hidden_img, mod_img + 2 * D, mod_img + 4 * D, mod_img + 3 * D,
                                           b->img_item, 0, xn_img, bRi};
      const omni_finish_adaln_stream ft = {Rt, Ri, L.to_add_out_b, hidden_txt, mod_txt + 2 * D, mod_txt + 4 * D, mod_txt + 3 * D,
                                           b->txt_item, 0, xn_txt, bRt};
      OMNI_TRY(omni_splitk_finish_adaln_pair(ws.splitk, ns, (int64_t)Ri + Rt, &fi, &ft, D, 6 * (int64_t)D, eps, stream));
      norm2_done = true;
    } else {
      OMNI_TRY(omni_gemm_bf16(&p, stream));
    }
  }
  // norm2 + modulate (reference :590, :595)
  const bool fused_q2 = f_up && blk;
  if (norm2_done) {
    // (performed by omni_splitk_finish_adaln_pair above)
  } else if (fused_q2) {
    const omni_adaln_stream si = {hidden_img, nullptr, Ri, mod_img + 4 * D, mod_img + 3 * D, b->img_item, 0, 0, ws.x8, Ri,
                                  ws.x8_scale};
    const omni_adaln_stream st = {hidden_txt, nullptr, Rt, mod_txt + 4 * D, mod_txt + 3 * D, b->txt_item, 0, 0,
                                  ws.x8 + (int64_t)Ri * D, Rt, ws.x8_scale + Ri};
    OMNI_TRY(adaln_streams(si, st, D, eps, stream));
  } else {
    const omni_adaln_stream si = {hidden_img, xn_img, Ri, mod_img + 4 * D, mod_img + 3 * D, b->img_item, 0, bRi, nullptr, 0,
                                  nullptr};
    const omni_adaln_stream st = {hidden_txt, xn_txt, Rt, mod_txt + 4 * D, mod_txt + 3 * D, b->txt_item, 0, bRt, nullptr, 0,
                                  nullptr};
    OMNI_TRY(adaln_streams(si, st, D, eps, stream));
  }
  // MLP up + GELU-tanh (reference :591, :596 -> diffusers FeedForward)
  {
    omni_gemm_params p = {};
    p.ngroups = 2; p.N = 4 * D; p.K = D; p.epilogue = OMNI_EPI_BIAS_GELU_TANH; p.w_k32_blocked = w->gemm_w_k32_blocked;
    p.g[0].a_k32_rows = bRi; p.g[1].a_k32_rows = bRt; p.g[0].out_k32_rows = bRi; p.g[1].out_k32_rows = bRt;
    p.g[0].A = xn_img; p.g[0].lda = D; p.g[0].M = Ri; p.g[0].W = L.img_mlp_w1; p.g[0].bias = L.img_mlp_b1;
    p.g[0].out = h_img; p.g[0].ldo = 4 * D;
    p.g[1].A = xn_txt; p.g[1].lda = D; p.g[1].M = Rt; p.g[1].W = L.txt_mlp_w1; p.g[1].bias = L.txt_mlp_b1;
    p.g[1].out = h_txt; p.g[1].ldo = 4 * D;
    p.g[0].tile_skip = pr.tile_img; p.g[1].tile_skip = pr.tile_txt;
    if (mlp_up_tail_split(Ri, Rt, 4 * D)) { p.splitk_ws = ws.splitk; p.splitk_ws_floats = ws.splitk_tail_floats; }
    if (f_up) {
      if (!fused_q2) OMNI_TRY(quant_streams(xn_img, bRi, xn_txt, bRt, D));
      fp8_streams(p, D, F->img_mlp_w1_8, F->img_mlp_w1_s, F->txt_mlp_w1_8, F->txt_mlp_w1_s);
    }
    OMNI_TRY(omni_gemm_bf16(&p, stream));
  }
  // MLP down + gated residual (reference :592, :597)
  {
    omni_gemm_params p = {};
    p.ngroups = 2; p.N = D; p.K = 4 * D; p.epilogue = OMNI_EPI_BIAS_GATE_RES; p.w_k32_blocked = w->gemm_w_k32_blocked;
    p.g[0].a_k32_rows = bRi; p.g[1].a_k32_rows = bRt;
    p.g[0].A = h_img; p.g[0].lda = 4 * D; p.g[0].M = Ri; p.g[0].W = L.img_mlp_w2; p.g[0].bias = L.img_mlp_b2;
    p.g[0].out = hidden_img; p.g[0].ldo = D; p.g[0].res = hidden_img; p.g[0].ldres = D;
    p.g[0].gate = mod_img + 5 * D; p.g[0].gate_item_stride = 6 * D; p.g[0].row_item_map = b->img_item;
    p.g[1].A = h_txt; p.g[1].lda = 4 * D; p.g[1].M = Rt; p.g[1].W = L.txt_mlp_w2; p.g[1].bias = L.txt_mlp_b2;
    p.g[1].out = hidden_txt; p.g[1].ldo = D; p.g[1].res = hidden_txt; p.g[1].ldres = D;
    p.g[1].gate = mod_txt + 5 * D; p.g[1].gate_item_stride = 6 * D; p.g[1].row_item_map = b->txt_item;
    p.g[0].tile_skip = pr.tile_img; p.g[1].tile_skip = pr.tile_txt;
    p.splitk_ws = ws.splitk; p.splitk_ws_floats = ws.splitk_floats;
    if (f_down) {
      OMNI_TRY(quant_streams(h_img, bRi, h_txt, bRt, 4 * D));
      fp8_streams(p, 4 * D, F->img_mlp_w2_8, F->img_mlp_w2_s, F->txt_mlp_w2_8, F->txt_mlp_w2_s);
    }
    // K-split launch inside omni_dit_forward's block loop, modulation rows from the request's table (the gate must outlive this
    // block): the NEXT block's norm1 finishes it
    const int ns = (pend && may_defer_down && fuse_ok && b->mod_table && (dit_fuse_finish() & 2)) ? omni_gemm_splitk_factor(&p) : 1;
    if (ns > 1) {
      p.kernel_hint = OMNI_GEMM_KERNEL_SPLITK_DEFER_FINISH;
      OMNI_TRY(omni_gemm_bf16(&p, stream));
      *pend = PendingFinish{ns, L.img_mlp_b2, L.txt_mlp_b2, mod_img + 5 * D, mod_txt + 5 * D};
    } else {
      OMNI_TRY(omni_gemm_bf16(&p, stream));
    }
  }
  return OMNI_OK;
}
struct NoHook { int operator()(omni_bf16*) const { return OMNI_OK; } };

// RoPE table row of every image / text stream row (joint_pos gathered through the joint-row maps): input of the fused
// q/k norm + RoPE epilogue
int prepare_positions(const omni_dit_batch* b, const Workspace& ws, omni_stream stream) {
  if (!dit_fuse_qkrope()) return OMNI_OK;
  OMNI_TRY(omni_internal_gather_i32(ws.img_pos, b->joint_pos, b->img_joint_row, b->n_img_rows, stream));
  OMNI_TRY(omni_internal_gather_i32(ws.txt_pos, b->joint_pos, b->txt_joint_row, b->n_txt_rows, stream));
  return OMNI_OK;
}
}  // namespace

extern "C" int omni_abi_version(void) { return 13; }
extern "C" const char* omni_build_arch(void) { return "gfx950"; }
extern "C" const char* omni_status_string(int status) {
  switch (status) {
    case OMNI_OK: return "ok";
    case OMNI_ERR_BAD_ARG: return "bad argument (null pointer or non-positive size)";
    case OMNI_ERR_UNSUPPORTED: return "unsupported shape for this kernel";
    case OMNI_ERR_LAUNCH: return "HIP launch failed";
    case OMNI_ERR_ALIGN: return "pointer or stride not sufficiently aligned";
    default: return "unknown status";
  }
}

extern "C" size_t omni_dit_workspace_bytes(const omni_dit_weights* w, int32_t n_img_rows, int32_t n_txt_rows,
                                           int32_t n_temb) {
  if (!w || n_img_rows <= 0 || n_txt_rows <= 0 || n_temb <= 0) return 0;
  return carve(nullptr, w, n_img_rows, n_txt_rows, n_temb).total;
}

extern "C" int omni_dit_forward(const omni_dit_weights* w, const omni_dit_batch* b, omni_stream stream) {
  if (!w || !b || !w->layers || !b->workspace) return OMNI_ERR_BAD_ARG;
  if (w->head_dim != 128 || w->in_channels % 64 != 0) return OMNI_ERR_UNSUPPORTED;
  const int32_t Ri = b->n_img_rows, Rt = b->n_txt_rows, nT = b->n_temb;
  const int32_t D = w->num_heads * w->head_dim;
  if (Ri <= 0 || Rt <= 0 || nT <= 0 || b->n_joint_rows != Ri + Rt) return OMNI_ERR_BAD_ARG;
  const Workspace ws = carve(b->workspace, w, Ri, Rt, nT);
  if (ws.total > b->workspace_bytes) return OMNI_ERR_BAD_ARG;
  const float eps = 1e-6f;

  // --- conditioning: sinusoid -> Linear -> SiLU -> Linear   (reference :50-62) ---------------------------
  OMNI_TRY(omni_timestep_sinusoid(b->timestep, nT, 256, 1000.0f, ws.tproj, stream));
  OMNI_TRY(linear_rows(ws.tproj, 256, nT, w->t_lin1_w, w->t_lin1_b, D, 256, ws.th, D, 0, 1, stream));
  OMNI_TRY(linear_rows(ws.th, D, nT, w->t_lin2_w, w->t_lin2_b, D, D, ws.temb, D, 0, 0, stream));
  // Layered variant: conditioning = timestep_emb + addition_t_embedding[additional_t_cond]   (reference :55-60)
  if (b->temb_add) OMNI_TRY(omni_internal_add_bf16(ws.temb, b->temb_add, (int64_t)nT * D, stream));

  // --- input projections (reference :743, :758-759) -------------------------------------------------------
  {
    omni_gemm_params p = {};
    p.ngroups = 1; p.N = D; p.K = w->in_channels; p.epilogue = OMNI_EPI_BIAS;
    p.g[0].A = b->latents; p.g[0].lda = w->in_channels; p.g[0].M = Ri;
    p.g[0].W = w->img_in_w; p.g[0].bias = w->img_in_b; p.g[0].out = ws.hidden_img; p.g[0].ldo = D;
    OMNI_TRY(omni_gemm_bf16(&p, stream));
  }
  OMNI_TRY(omni_rmsnorm(b->prompt_embeds, w->joint_dim, ws.txt_normed, w->joint_dim, Rt, w->joint_dim,
                        w->txt_norm_w, eps, stream));
  {
    omni_gemm_params p = {};
    p.ngroups = 1; p.N = D; p.K = w->joint_dim; p.epilogue = OMNI_EPI_BIAS;
    p.g[0].A = ws.txt_normed; p.g[0].lda = w->joint_dim; p.g[0].M = Rt;
    p.g[0].W = w->txt_in_w; p.g[0].bias = w->txt_in_b; p.g[0].out = ws.hidden_txt; p.g[0].ldo = D;
    p.splitk_ws = ws.splitk; p.splitk_ws_floats = ws.splitk_floats;
    OMNI_TRY(omni_gemm_bf16(&p, stream));
  }

  OMNI_TRY(prepare_positions(b, ws, stream));
  const omni_teacache* tc = b->teacache;
  const int32_t rows_per_item = b->n_items > 0 ? Ri / b->n_items : 0;
  if (tc) {
    if (rows_per_item * b->n_items != Ri || b->n_items > 64) return OMNI_ERR_UNSUPPORTED;
    // the image stream as it enters the block stack: residual = out - this; skipped items leave as this + cached residual
    if (hipMemcpyAsync(ws.h_in, ws.hidden_img, (size_t)Ri * D * sizeof(omni_bf16), hipMemcpyDeviceToDevice,
                       static_cast<hipStream_t>(stream)) != hipSuccess)
      return OMNI_ERR_LAUNCH;
  }
  const BlockPred pred = tc ? BlockPred{tc->tile_skip_img, tc->tile_skip_txt, tc->skip} : BlockPred{nullptr, nullptr, nullptr};
  const int32_t blocked = (dit_act_blocked() && (D % 32 == 0)) ? 1 : 0;
  PendingFinish pend;                                   // block l's MLP-down finish, performed by block l + 1's norm1
  for (int l = 0; l < w->num_layers; ++l) {
    const bool more = l + 1 < w->num_layers;
    if (l == 0 && tc) {
      // decision on the first block's modulated input, BEFORE anything of the block stack that could be skipped; the
      // predicates it writes gate every GEMM row tile / attention block of skipped items from here on
      auto hook = [&](omni_bf16* xn_img) {
        return omni_internal_teacache_decide(tc, xn_img, b->n_items, rows_per_item, Ri, Rt, D, blocked, stream);
      };
      OMNI_TRY(run_block(w, l, b, ws, ws.hidden_img, ws.hidden_txt, ws.temb, pred, hook, stream, BLOCK_ALL, nullptr, &pend, more));
    } else {
      OMNI_TRY(run_block(w, l, b, ws, ws.hidden_img, ws.hidden_txt, ws.temb, pred, NoHook{}, stream, BLOCK_ALL, nullptr, &pend, more));
    }
  }
  if (pend.nsplit > 1) return OMNI_ERR_LAUNCH;          // (cannot happen: the last block never defers)
  if (tc) OMNI_TRY(omni_internal_teacache_post(tc, ws.hidden_img, ws.h_in, Ri, rows_per_item, D, stream));

  omni_bf16* xn_img = ws.xn;
  // --- output head: AdaLayerNormContinuous (scale first, then shift) + proj_out (reference :797-798) -------
  OMNI_TRY(linear_rows(ws.temb, D, nT, w->norm_out_w, w->norm_out_b, 2 * (int64_t)D, D, ws.emb_out, 2 * D, 1,
                                  0, stream));
  OMNI_TRY(omni_adaln_modulate(ws.hidden_img, D, xn_img, D, Ri, D, ws.emb_out, ws.emb_out + D, 2 * D, b->img_item, 0,
                               eps, stream));
  {
    omni_gemm_params p = {};
    p.ngroups = 1; p.N = w->out_channels_packed; p.K = D; p.epilogue = OMNI_EPI_BIAS;
    p.g[0].A = xn_img; p.g[0].lda = D; p.g[0].M = Ri; p.g[0].W = w->proj_out_w; p.g[0].bias = w->proj_out_b;
    p.g[0].out = b->noise_pred; p.g[0].ldo = w->out_channels_packed;
    p.splitk_ws = ws.splitk; p.splitk_ws_floats = ws.splitk_floats;
    OMNI_TRY(omni_gemm_bf16(&p, stream));
  }
  return OMNI_OK;
}

extern "C" size_t omni_dit_modulation_table_workspace_bytes(const omni_dit_weights* w, int32_t M) {
  if (!w || M <= 0) return 0;
  const int64_t D = (int64_t)w->num_heads * w->head_dim;
  return align_up((size_t)M * D * sizeof(omni_bf16)) + align_up((size_t)8 * M * 6 * D * sizeof(float));
}

extern "C" int omni_dit_modulation_table(const omni_dit_weights* w, const omni_bf16* temb, int32_t M, omni_bf16* table,
                                         void* workspace, size_t workspace_bytes, omni_stream stream) {
  if (!w || !w->layers || !temb || !table || !workspace || M <= 0) return OMNI_ERR_BAD_ARG;
  const int64_t D = (int64_t)w->num_heads * w->head_dim;
  if (D % 64 != 0) return OMNI_ERR_UNSUPPORTED;
  if (workspace_bytes < omni_dit_modulation_table_workspace_bytes(w, M)) return OMNI_ERR_BAD_ARG;
  if (M <= 8) {
    // a handful of rows (BASELINE config 1: 4 steps): the weight-streaming GEMV, one launch per matrix at HBM speed (the GEMM
    // kernel below needs two launches per matrix and is paced by one CU's k-loop per 256-column panel).  Measured, 60 layers
    // (profiles/r06e_table_pass.txt): GEMV 2.5 / 3.0 / 3.0 / 3.4 / 4.3 / 4.2 ms at M = 1 .. 6 (5.4 .. 3.3 TB/s), 7.1 / 7.9 ms at
    // M = 7 / 8 (8 staged fp32 rows are 96 KB of LDS: one block per CU); the GEMM path: 4.3 .. 4.9 ms at M = 7 .. 50.  M = 7 / 8
    // stay on the GEMV nevertheless: the two paths round SiLU differently (fp32 inside the GEMV, bf16 in front of the GEMM as the
    // reference's eager Sequential does), and a table must agree with the per-forward GEMVs of the module-level plug-in path
    // (TeaCache's host hook walks that one) to the GEMV's own bits wherever both exist — tests/test_gpu_teacache.py; a schedule's
    // table is computed once and kept (modulation_table_for_schedule), so the 3 ms are paid once per schedule
    for (int l = 0; l < w->num_layers; ++l) {
      const omni_dit_layer_weights& L = w->layers[l];
      OMNI_TRY(omni_linear_smallbatch(temb, D, M, L.img_mod_w, L.img_mod_b, 6 * D, (int32_t)D,
                                      table + (int64_t)(2 * l + 0) * M * 6 * D, 6 * D, 1, 0, stream));
      OMNI_TRY(omni_linear_smallbatch(temb, D, M, L.txt_mod_w, L.txt_mod_b, 6 * D, (int32_t)D,
                                      table + (int64_t)(2 * l + 1) * M * 6 * D, 6 * D, 1, 0, stream));
    }
    return OMNI_OK;
  }
  omni_bf16* act = static_cast<omni_bf16*>(workspace);                        // silu(temb), rounded to bf16 as the reference's
  float* splitk = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)M * D * sizeof(omni_bf16)));
  OMNI_TRY(omni_internal_silu_bf16(act, temb, (int64_t)M * D, stream));      // nn.SiLU in front of the Linear (:478-481)
  for (int l = 0; l < w->num_layers; ++l) {
    const omni_dit_layer_weights& L = w->layers[l];
    for (int s = 0; s < 2; ++s) {
      omni_gemm_params p = {};
      p.ngroups = 1; p.N = (int32_t)(6 * D); p.K = (int32_t)D; p.epilogue = OMNI_EPI_BIAS;
      p.g[0].A = act; p.g[0].lda = D; p.g[0].M = M;
      p.g[0].W = s ? L.txt_mod_w : L.img_mod_w; p.g[0].bias = s ? L.txt_mod_b : L.img_mod_b;
      p.g[0].out = table + (int64_t)(2 * l + s) * M * 6 * D; p.g[0].ldo = 6 * D;
      p.splitk_ws = splitk; p.splitk_ws_floats = (int64_t)8 * M * 6 * D;
      OMNI_TRY(omni_gemm_bf16(&p, stream));
    }
  }
  return OMNI_OK;
}

extern "C" int omni_dit_block(const omni_dit_weights* w, int32_t layer, const omni_dit_batch* b, omni_bf16* hidden_img,
                              omni_bf16* hidden_txt, const omni_bf16* temb, omni_stream stream) {
  if (!w || !b || !w->layers || !b->workspace || !hidden_img || !hidden_txt || !temb) return OMNI_ERR_BAD_ARG;
  if (layer < 0 || layer >= w->num_layers) return OMNI_ERR_BAD_ARG;
  if (w->head_dim != 128) return OMNI_ERR_UNSUPPORTED;
  const int32_t Ri = b->n_img_rows, Rt = b->n_txt_rows, nT = b->n_temb;
  if (Ri <= 0 || Rt <= 0 || nT <= 0 || b->n_joint_rows != Ri + Rt) return OMNI_ERR_BAD_ARG;
  const Workspace ws = carve(b->workspace, w, Ri, Rt, nT);
  if (ws.total > b->workspace_bytes) return OMNI_ERR_BAD_ARG;
  OMNI_TRY(prepare_positions(b, ws, stream));
  return run_block(w, layer, b, ws, hidden_img, hidden_txt, temb, BlockPred{nullptr, nullptr, nullptr}, NoHook{}, stream);
}

extern "C" int omni_dit_block_qkv(const omni_dit_weights* w, int32_t layer, const omni_dit_batch* b, omni_bf16* hidden_img,
                                  omni_bf16* hidden_txt, const omni_bf16* temb, omni_bf16** q, omni_bf16** k, omni_bf16** v,
                                  omni_stream stream) {
  if (!w || !b || !w->layers || !b->workspace || !hidden_img || !hidden_txt || !temb || !q || !k || !v) return OMNI_ERR_BAD_ARG;
  if (layer < 0 || layer >= w->num_layers) return OMNI_ERR_BAD_ARG;
  if (w->head_dim != 128) return OMNI_ERR_UNSUPPORTED;
  const int32_t Ri = b->n_img_rows, Rt = b->n_txt_rows, nT = b->n_temb;
  if (Ri <= 0 || Rt <= 0 || nT <= 0 || b->n_joint_rows != Ri + Rt) return OMNI_ERR_BAD_ARG;
  const Workspace ws = carve(b->workspace, w, Ri, Rt, nT);
  if (ws.total > b->workspace_bytes) return OMNI_ERR_BAD_ARG;
  OMNI_TRY(prepare_positions(b, ws, stream));
  *q = ws.q; *k = ws.k; *v = ws.v;
  return run_block(w, layer, b, ws, hidden_img, hidden_txt, temb, BlockPred{nullptr, nullptr, nullptr}, NoHook{}, stream,
                   BLOCK_QKV);
}

extern "C" int omni_dit_block_post(const omni_dit_weights* w, int32_t layer, const omni_dit_batch* b, omni_bf16* hidden_img,
                                   omni_bf16* hidden_txt, const omni_bf16* temb, const omni_bf16* attn, omni_stream stream) {
  if (!w || !b || !w->layers || !b->workspace || !hidden_img || !hidden_txt || !temb || !attn) return OMNI_ERR_BAD_ARG;
  if (layer < 0 || layer >= w->num_layers) return OMNI_ERR_BAD_ARG;
  if (!omni_aligned16(attn)) return OMNI_ERR_ALIGN;
  const int32_t Ri = b->n_img_rows, Rt = b->n_txt_rows, nT = b->n_temb;
  if (Ri <= 0 || Rt <= 0 || nT <= 0 || b->n_joint_rows != Ri + Rt) return OMNI_ERR_BAD_ARG;
  const Workspace ws = carve(b->workspace, w, Ri, Rt, nT);
  if (ws.total > b->workspace_bytes) return OMNI_ERR_BAD_ARG;
  return run_block(w, layer, b, ws, hidden_img, hidden_txt, temb, BlockPred{nullptr, nullptr, nullptr}, NoHook{}, stream,
                   BLOCK_POST, attn);
}
