/*
 * omni_cdna4.h — C ABI of libomni_cdna4.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * Qwen-Image DiT denoising hot path of vllm-omni.
 *
 * The reference (vllm-project/vllm-omni) is 100 % Python and has NO FFI for this path; every GPU
 * kernel it runs comes from PyTorch / vLLM / flash-attn / diffusers.  The entry points below are
 * what a binding for its L0 plug-in points would call.  Each one cites the
 * reference call site (path:line relative to the reference root) whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, explicit sizes/strides, a hipStream_t passed as void*;
 *     no torch / ATen types, no allocation inside any call (the caller owns every buffer).
 *   - bf16 tensors are `uint16_t` bit patterns (storage bf16, all arithmetic fp32, one rounding).
 *   - every function returns OMNI_OK (0) or a negative omni_status; it never throws or aborts.
 *     Launch errors are reported through hipGetLastError() -> OMNI_ERR_LAUNCH.
 *   - thread-safe for concurrent calls on different streams; no global mutable state.
 *   - "rows" are tokens.  Batches are RAGGED: B items are concatenated along the row axis
 *     and described by `cu_seqlens` (B+1 prefix sums) or per-row int32 maps, so step-batched
 *     requests with different text lengths need no padding.
 */
#ifndef OMNI_CDNA4_H
#define OMNI_CDNA4_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared in this header are exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef uint16_t omni_bf16;
typedef void* omni_stream; /* hipStream_t */

typedef enum {
  OMNI_OK = 0,
  OMNI_ERR_BAD_ARG = -1,     /* null pointer, non-positive size                       */
  OMNI_ERR_UNSUPPORTED = -2, /* shape outside what the kernel is built for (see each) */
  OMNI_ERR_LAUNCH = -3,      /* hipGetLastError() != hipSuccess after the launch      */
  OMNI_ERR_ALIGN = -4        /* pointer / stride not 16-byte aligned where required   */
} omni_status;

/* Library identity: ABI version (bumped on any signature change) and the gfx target it was built for. */
int omni_abi_version(void);
const char* omni_build_arch(void);
const char* omni_status_string(int status);

/* ------------------------------------------------------------------------------------------------
 * Grouped GEMM with fused epilogue:  Y_g = epilogue(A_g · W_gᵀ + bias_g)  for g < ngroups (<= 2).
 * Replaces the hipBLASLt GEMMs behind F.linear at
 *   vllm_omni/diffusion/models/qwen_image/qwen_image_transformer.py:380,384 (to_qkv / add_kv_proj),
 *   :452,456 (to_out[0] / to_add_out), :491,501,591,596 (FeedForward), :743,759,798 (img_in/txt_in/proj_out)
 * plus the elementwise ops the reference runs after them (:586-587,592,597 gated residuals,
 * GELU-tanh inside FeedForward, torch.cat at :414-416 for the QKV split-to-joint write).
 * Both groups (image stream, text stream) share N, K and the epilogue and run in ONE launch.
 *
 * A_g  [M_g, K] bf16 row-major, row stride lda (elements); optional a_row_map[M_g] gathers rows.
 * W_g  [N, K]   bf16 row-major ([out, in], the nn.Linear layout; fused QKV rows ordered q|k|v).
 * K % 64 == 0 (else OMNI_ERR_UNSUPPORTED); M_g, N arbitrary (tails are predicated).
 * ---------------------------------------------------------------------------------------------- */
typedef enum {
  OMNI_EPI_BIAS = 0,          /* y = acc + bias                                              */
  OMNI_EPI_BIAS_GELU_TANH = 1,/* y = gelu_tanh(acc + bias)                                   */
  OMNI_EPI_BIAS_GATE_RES = 2, /* y = res + gate[item(row)] * (acc + bias)   (res may alias y) */
  OMNI_EPI_BIAS_SPLIT3 = 3,   /* y = acc + bias, column block n/split_n selects out/out1/out2 */
  OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE = 4 /* SPLIT3, and the out / out1 blocks (q, k) additionally get the per-head
                               * RMSNorm(128) * w and the interleaved RoPE of omni_qk_norm_rope applied to the bf16-rounded
                               * linear output before the store (bit-identical to SPLIT3 followed by omni_qk_norm_rope);
                               * needs split_n % 128 == 0 and the qk_* fields of every group */
} omni_epilogue;

typedef struct {
  const omni_bf16* A;
  int64_t lda;
  const int32_t* a_row_map; /* nullable: source row of logical row r (gather)              */
  int32_t M;
  const omni_bf16* W;       /* [N, K], ld = K                                              */
  const omni_bf16* bias;    /* [N], nullable                                               */
  omni_bf16* out;           /* [*, ldo]                                                    */
  omni_bf16* out1;          /* SPLIT3 only: columns [split_n, 2*split_n)                   */
  omni_bf16* out2;          /* SPLIT3 only: columns [2*split_n, 3*split_n)                 */
  int64_t ldo;
  const int32_t* out_row_map; /* nullable: destination row of logical row r (scatter)      */
  const omni_bf16* res;     /* GATE_RES: residual, same row indexing as out                */
  int64_t ldres;
  const omni_bf16* gate;    /* GATE_RES: gate[item * gate_item_stride + n]                 */
  int64_t gate_item_stride;
  const int32_t* row_item_map; /* nullable: item (batch element) of logical row r          */
  int32_t rows_per_item;    /* used when row_item_map == NULL: item = r / rows_per_item    */
  /* SPLIT3_QKNORM_ROPE only: RMSNorm weights [128] of the q and k heads of this stream, RoPE tables [npos, 64] and the
   * table row of every OUTPUT row (indexed by out_row_map[r], or r) */
  const omni_bf16* qk_norm_q_w;
  const omni_bf16* qk_norm_k_w;
  const omni_bf16* qk_rope_cos;
  const omni_bf16* qk_rope_sin;
  const int32_t* qk_row_pos;
  float qk_eps;
  int32_t a_k32_rows;       /* 0: A is row-major [*, lda].  R > 0: A is K32-blocked [K/32][R][32] (element (r,k) at
                             * ((k/32)*R + r)*32 + k%32; lda ignored; a_row_map still picks r).  Same idea as
                             * omni_gemm_params.w_k32_blocked, for activations produced by this library's own kernels */
  int32_t out_k32_rows;     /* 0: out is row-major [*, ldo].  R > 0: out is written K32-blocked [N/32][R][32] (ldo
                             * ignored; BIAS / BIAS_GELU_TANH epilogues only) so that the next GEMM can read it blocked */
  float qk_q_scale;         /* SPLIT3_QKNORM_ROPE only, ABI v5 (occupies former padding): 0 or 1 = off; otherwise the q block is
                             * additionally multiplied by this factor in fp32 BEFORE its single bf16 rounding (folded into the
                             * RMSNorm weight).  omni_dit_forward passes softmax_scale * log2(e), so that the attention kernel's
                             * QK^T accumulator is the exp2 argument directly (csrc/attention.hip OMNI_ATTN_BAKE) */
  const int32_t* tile_skip; /* nullable DEVICE array, one int32 per 256-row tile of this group: non-zero = the workgroups of
                             * that row tile return at once (its rows belong to items whose block stack is skipped this
                             * forward, omni_teacache).  A device-side predicate: no host round trip.  ABI v3. */
  /* ABI v7 — omni_gemm_params.fp8 only: fp32 dequantisation scales, a_scale[r] of A's STORED row r (the row a_row_map picks,
   * or the logical row) and w_scale[n] of output channel n:  Y = epilogue((A8 . W8^T) * a_scale[r] * w_scale[n] + bias[n]).
   * Both come from omni_quantize_fp8_rows. */
  const float* a_scale;
  const float* w_scale;
} omni_gemm_group;

typedef struct {
  int32_t ngroups; /* 1 or 2 */
  int32_t N, K;
  int32_t epilogue; /* omni_epilogue */
  int32_t split_n;  /* SPLIT3: width of each output (multiple of 32) */
  int32_t w_k32_blocked; /* 0: W is [N, K] row-major (the reference's nn.Linear.weight).  1: W is the SAME values in
                          * K32-blocked order [K/32][N][32] (element (n,k) at ((k/32)*N + n)*32 + k%32): a 32-wide k-slab of
                          * 16 consecutive rows is then 1 KiB contiguous, so every LDS-DMA piece of the weight operand
                          * fetches whole 128-B lines instead of 16 half lines (the L1 request-slot limit).
                          * A one-time re-layout at weight-load time; occupies the struct's former padding. */
  omni_gemm_group g[2];
  /* ABI v4 — optional split-K workspace (DEVICE, 16-byte aligned, caller-owned; NULL = never split).  When the launch has so
   * few tiles that at least half of the CUs would idle (<= 128 tiles in <= 10 row tiles: a forward over one or two small images
   * streams its N = 3072 weight panels through 36 workgroups), the K loop is split s ways (s <= 8), the fp32 partial tiles go to
   * splitk_ws[s][M0 + M1][N] and a second kernel sums them in split order and runs the epilogue.  Needs
   * s * (M0 + M1) * N <= splitk_ws_floats; results agree with the unsplit kernel to fp32 summation order
   * (ABI v12: OMNI_GEMM_KERNEL_SPLITK_IN_LAUNCH below moves the reduction into the launch.) */
  float* splitk_ws;
  int64_t splitk_ws_floats;
  /* ABI v6 — kernel choice, normally 0 = automatic (the ping-pong kernel where its preconditions hold, else the ring kernel).
   * OMNI_GEMM_KERNEL_RING forces the ring kernel (the fallback family): the two accumulate every output element in the same
   * k order and must agree bit for bit, which is how callers (and tests/) cross-check one against the other.  Values
   * >= 16 select development families and are honoured only by libraries built with -DOMNI_DEV; the product library ignores
   * them.  The library itself has no switches: no environment variable and no process-global setter changes what a call does. */
  int32_t kernel_hint;
  /* ABI v7 — fp8 = 1: A and W hold OCP fp8 e4m3 values (one byte each) instead of bf16, in the K64-blocked order
   * [K/64][rows][64] (what omni_quantize_fp8_rows writes; a_k32_rows / w_k32_blocked then carry the row counts R / 1 as for
   * bf16 — byte for byte it is the K32-blocked layout of a bf16 matrix with K/2 columns), with per-row / per-output-channel
   * fp32 scales in omni_gemm_group.a_scale / w_scale.  K counts fp8 elements and must be a multiple of 128.  The products run
   * on v_mfma_scale_f32_16x16x128_f8f6f4 (twice the bf16 MFMA rate; block scales fixed at 1, the fp32 scales are applied
   * to the fp32 accumulators in the epilogue).  No split-K.  Outputs, bias and every epilogue stay bf16 as without it. */
  int32_t fp8;
} omni_gemm_params;
#define OMNI_GEMM_KERNEL_AUTO 0
#define OMNI_GEMM_KERNEL_RING 1
/* ABI v10 — automatic kernel choice, and the split-K rule above without its "<= 10 row tiles" clause: for a TALL launch with a
 * long K and few tiles (the VAE mid-block attention's P.V: 8192 x 384 x 16384 = 64 tiles).  The clause exists so that a batch
 * and its shards take the same decision; a caller that passes this vouches that nothing compares the result across batch
 * compositions bit for bit. */
#define OMNI_GEMM_KERNEL_SPLITK_TALL 2
/* ABI v11 — TAIL SPLIT (automatic with a split-K workspace, this hint turns it off).  A launch of more than one round of 256x256
 * tiles whose last round is thin (tiles % CUs at most a quarter of the CUs: 1548 tiles on 256 CUs = 6 rounds + 12 tiles, the
 * N = 3072 GEMMs of one 2048x2048 request) runs the tiles of the full rounds unsplit — bit-identical to a launch without workspace — and splits the K
 * loop of the remaining tiles s ways (s = 4 or 8: 12 / 24 K-tiles per piece; fp32 partials splitk_ws[tail tile][s][256][256] as far
 * as splitk_ws_floats allows — up to 64 x 8 x 65536 floats — a small second kernel runs their epilogue), so the partial round
 * costs its share of the work instead of a whole tile time.  The tail tiles then agree with the unsplit
 * kernel to fp32 summation order; WHICH tiles are tail tiles depends on the launch's tile count, i.e. on the batch composition —
 * a caller that compares results bit for bit across batch compositions passes OMNI_GEMM_KERNEL_NO_TAIL_SPLIT (or no workspace). */
#define OMNI_GEMM_KERNEL_NO_TAIL_SPLIT 3
/* ABI v12 — split-K (the ABI v4 rule) with the reduction INSIDE the launch, opt-in: needs 512 floats behind the partials
 * (s * (M0 + M1) * N + 512 <= splitk_ws_floats) and the grid 8 * ceil(tiles / 8) * s within one round of the CUs, else the
 * two-kernel path runs.  The s workgroups of a tile (all resident at once, placed on one XCD) publish their fp32 partials
 * write-through, meet at an arrival counter and each runs the epilogue for its share of the tile's rows — no second kernel, the
 * same bits (the sum is formed in split order by the same code).  The last 512 words of the workspace hold the counters: the
 * call zeroes them on the stream before its launch and the launch leaves them zero; a wait that cannot complete (a device
 * whose usable CUs are fewer than it reports) gives up after about a second, sets word 256 and returns wrong rows rather than
 * hanging the device.  Measured on MI355X it pays only for 2-way splits on grids that nearly fill the chip (+1.1 .. 1.3 % per
 * forward) and loses for deep splits: not the default. */
#define OMNI_GEMM_KERNEL_SPLITK_IN_LAUNCH 4
/* ABI v13 — split-K (the ABI v4 rule) WITHOUT its finish kernel: the call launches only the K-split main kernel and leaves the
 * fp32 partials splitk_ws[s][M0 + M1][N] (group 1's rows behind group 0's; no bias, no epilogue applied) for a consumer that
 * folds the reduction into its own pass — omni_splitk_finish_adaln_pair below for OMNI_EPI_BIAS_GATE_RES.  The caller asks
 * omni_gemm_splitk_factor first: the hint is valid only for a call whose factor is > 1 (anything else returns
 * OMNI_ERR_UNSUPPORTED — the library never silently runs a different path).  The output / residual / gate / bias fields of the
 * groups are not touched by such a call. */
#define OMNI_GEMM_KERNEL_SPLITK_DEFER_FINISH 5

int omni_gemm_bf16(const omni_gemm_params* p, omni_stream stream);
/* ABI v13 — the whole-launch split-K factor omni_gemm_bf16 would use for `p` on the current device (a function of the tile counts,
 * K and the workspace size only): 1 = the call does not split, 0 = `p` is not a valid call. */
int omni_gemm_splitk_factor(const omni_gemm_params* p);

/* ABI v7 — dynamic per-row fp8 (OCP e4m3) quantisation of a bf16 matrix for omni_gemm_params.fp8:
 *   scale[r] = max(amax_k |x[r, k]|, tiny) / 448;   y8[r, k] = e4m3_rn(x[r, k] / scale[r])
 * x: [rows, K] bf16, row-major with stride ldx, or K32-blocked [K/32][x_k32_rows][32] when x_k32_rows > 0 (ldx ignored).
 * y8: K64-blocked bytes [K/64][y_rows][64] (y_rows >= rows: the row count of the buffer, = omni_gemm_group.a_k32_rows of
 * the consuming GEMM); scale: fp32 [rows].  K % 64 == 0, K <= 16384.  One wave per row, the row held in registers: x is
 * read once (2 B/element) and y8 written once (1 B/element): HBM-bound.  Used for activations (per token) and, once at
 * load time, for weights (per output channel).  (vLLM's dynamic per-token fp8 recipe; the reference has no fp8 path —
 * BASELINE.json config 5 asks for one.) */
int omni_quantize_fp8_rows(const omni_bf16* x, int64_t ldx, int32_t x_k32_rows, int32_t rows, int32_t K, uint8_t* y8,
                           int32_t y_rows, float* scale, omni_stream stream);

/* ------------------------------------------------------------------------------------------------
 * AdaLN-modulate:  y = LayerNorm(x; eps, no affine) * (1 + scale[item]) + shift[item]
 * Replaces AdaLayerNorm.forward_hip -> forward_native, vllm_omni/diffusion/layers/adalayernorm.py:70-76,94-102
 * (nn.LayerNorm + mul + add = 3 passes -> 1), and the LayerNorm half of diffusers
 * AdaLayerNormContinuous (call site qwen_image_transformer.py:797).
 * x,y [rows, D] bf16 (row strides ldx/ldy); scale/shift bf16 vectors of length D per item with
 * stride mod_item_stride; item(row) = row_item_map ? row_item_map[row] : row / rows_per_item.
 * D % 8 == 0 and D <= 8192.
 * ---------------------------------------------------------------------------------------------- */
int omni_adaln_modulate(const omni_bf16* x, int64_t ldx, omni_bf16* y, int64_t ldy, int32_t rows, int32_t D,
                        const omni_bf16* scale, const omni_bf16* shift, int64_t mod_item_stride,
                        const int32_t* row_item_map, int32_t rows_per_item, float eps, omni_stream stream);
/* Same, with the output optionally K32-blocked: y_k32_rows = 0 -> row-major (ldy); R > 0 -> y is [D/32][R][32]
 * (omni_gemm_group.a_k32_rows of the GEMM that consumes it; D % 32 == 0; ldy ignored). */
int omni_adaln_modulate_ex(const omni_bf16* x, int64_t ldx, omni_bf16* y, int64_t ldy, int32_t rows, int32_t D,
                           const omni_bf16* scale, const omni_bf16* shift, int64_t mod_item_stride,
                           const int32_t* row_item_map, int32_t rows_per_item, float eps, int32_t y_k32_rows,
                           omni_stream stream);

/* ABI v7 — AdaLN-modulate with the result quantised to fp8 in the same pass (the row is in registers anyway):
 * y8 / y8_scale as omni_quantize_fp8_rows would produce them from the bf16-ROUNDED result (bit-identical to running the two
 * kernels one after the other), K64-blocked with y8_rows rows.  y (nullable): additionally the bf16 result, K32-blocked with
 * y_k32_rows rows (TeaCache reads the modulated input of the first block).  D % 64 == 0. */
int omni_adaln_modulate_fp8(const omni_bf16* x, int64_t ldx, int32_t rows, int32_t D, const omni_bf16* scale,
                            const omni_bf16* shift, int64_t mod_item_stride, const int32_t* row_item_map,
                            int32_t rows_per_item, float eps, omni_bf16* y, int32_t y_k32_rows, uint8_t* y8,
                            int32_t y8_rows, float* y8_scale, omni_stream stream);

/* ABI v12 — AdaLN-modulate of TWO row groups in ONE launch: the image stream's rows and the text stream's rows of a DiT block
 * (reference qwen_image_transformer.py:564-567 and :590,:595 run `img_norm1` / `txt_norm1`, `img_norm2` / `txt_norm2` back to
 * back).  Each group is what omni_adaln_modulate_ex / omni_adaln_modulate_fp8 take for one call — x row-major [rows, D] with
 * row stride D; y (nullable when y8 is given) row-major with row stride D, or K32-blocked with y_k32_rows rows; y8 / y8_scale
 * (nullable pair) the fp8 copy, K64-blocked with y8_rows rows — and gets the same bits; D, eps and the modulation stride are
 * shared.  At small batches (one 256x256 CFG pair: 512 + 128 rows) the two launches are latency, not bandwidth. */
typedef struct omni_adaln_stream {
  const omni_bf16* x;
  omni_bf16* y;
  int32_t rows;
  const omni_bf16* scale;
  const omni_bf16* shift;
  const int32_t* row_item_map;
  int32_t rows_per_item;
  int32_t y_k32_rows;
  uint8_t* y8;
  int32_t y8_rows;
  float* y8_scale;
} omni_adaln_stream;
int omni_adaln_modulate_pair(const omni_adaln_stream* a, const omni_adaln_stream* b, int32_t D, int64_t mod_item_stride,
                             float eps, omni_stream stream);

/* ABI v13 — the finish of a DEFERRED split-K GEMM with the gated-residual epilogue (omni_gemm_bf16 with
 * OMNI_GEMM_KERNEL_SPLITK_DEFER_FINISH, epilogue OMNI_EPI_BIAS_GATE_RES, N = D) fused with the AdaLN-modulate that follows it in a
 * DiT block: the attention output projection -> norm2 (qwen_image_transformer.py:586-590,595) and the MLP down projection -> the
 * NEXT block's norm1 (:592,597 -> :564-567).  One wave per row, both streams of the block in ONE launch:
 *   c      = bf16(sum_s partial[s][row] + bias)                  (split order, the finish kernel's arithmetic)
 *   hidden = bf16(hidden + gate[item(row)] * c)                  (written back in place: the residual stream)
 *   y      = bf16(LN(hidden) * (1 + scale[item]) + shift[item])   (omni_adaln_modulate_ex's arithmetic on the rounded row)
 * — the same bits as the finish kernel followed by omni_adaln_modulate_pair, without the second pass over the residual stream,
 * two of the three launches and their boundaries (at one 256x256 CFG pair the three are latency, not bandwidth).
 * Per stream: `rows` rows, `ws_row0` = its first row inside the partials' M0 + M1 rows; hidden row-major [rows, D] (row stride D);
 * y row-major (row stride D) or K32-blocked with y_k32_rows rows; gate / scale / shift rows of mod_item_stride elements indexed by
 * row_item_map[row] (or row / rows_per_item).  nsplit in {2, 3, 4, 6, 8}; D % 8 == 0, D <= 4096. */
typedef struct omni_finish_adaln_stream {
  int32_t rows;
  int32_t ws_row0;
  const omni_bf16* bias; /* [D], nullable */
  omni_bf16* hidden;
  const omni_bf16* gate;
  const omni_bf16* scale;
  const omni_bf16* shift;
  const int32_t* row_item_map;
  int32_t rows_per_item;
  omni_bf16* y;
  int32_t y_k32_rows;
} omni_finish_adaln_stream;
int omni_splitk_finish_adaln_pair(const float* splitk_ws, int32_t nsplit, int64_t ws_rows, const omni_finish_adaln_stream* a,
                                  const omni_finish_adaln_stream* b, int32_t D, int64_t mod_item_stride, float eps,
                                  omni_stream stream);

/* RMSNorm over the last dim with learned weight: y = x * rsqrt(mean(x^2) + eps) * w.
 * Replaces vllm RMSNorm at qwen_image_transformer.py:758 (txt_norm, D = 3584).  D % 8 == 0, D <= 8192. */
int omni_rmsnorm(const omni_bf16* x, int64_t ldx, omni_bf16* y, int64_t ldy, int32_t rows, int32_t D,
                 const omni_bf16* weight, float eps, omni_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Per-head RMSNorm (head_dim 128, learned weight) + interleaved RoPE, in place on a [rows, H*128] tensor.
 * Replaces qwen_image_transformer.py:397-400 (vllm RMSNorm on q,k) and :403-410
 * (RotaryEmbedding(is_neox_style=False), vllm_omni/diffusion/layers/rope.py:12-36,108-127):
 *   xn = x * rsqrt(mean_128(x^2) + eps) * w[stream(row)]
 *   out[2i] = xn[2i]*cos[p,i] - xn[2i+1]*sin[p,i] ; out[2i+1] = xn[2i+1]*cos[p,i] + xn[2i]*sin[p,i]
 * cos/sin: bf16 tables [npos, 64] (the reference casts them to the activation dtype first, :403-406).
 * row_pos[row] (int32) = table row; weight = w_txt if row_pos[row] < txt_pos_end else w_img.
 * ---------------------------------------------------------------------------------------------- */
int omni_qk_norm_rope(omni_bf16* x, int64_t ldx, int32_t rows, int32_t num_heads,
                      const omni_bf16* w_img, const omni_bf16* w_txt, const omni_bf16* cos_tab,
                      const omni_bf16* sin_tab, const int32_t* row_pos, int32_t txt_pos_end, float eps,
                      omni_stream stream);

/* Standalone interleaved RoPE on [B, S, H, dh] (the RotaryEmbedding.forward_hip plug-in point,
 * vllm_omni/diffusion/layers/rope.py:108-127).  cos/sin bf16 [S, dh/2]; x -> y (may alias). dh % 16 == 0. */
int omni_rope_interleaved(const omni_bf16* x, omni_bf16* y, int32_t B, int32_t S, int32_t H, int32_t dh,
                          const omni_bf16* cos_tab, const omni_bf16* sin_tab, omni_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Flash attention forward, non-causal, no mask, head_dim 128, bf16 in/out, fp32 softmax/accumulate.
 * Replaces F.scaled_dot_product_attention at vllm_omni/diffusion/attention/backends/sdpa.py:46-66
 * (and flash_attn_func / sageattn in the sibling backends).
 * q,k,v,out: [total_rows, H*128] with row strides ld* (the reference's [B,S,H,dh] "NHD" layout
 * flattened over B,S); item b owns rows [cu_seqlens[b], cu_seqlens[b+1]).  cu_seqlens is a
 * DEVICE int32 array of B+1 entries.  max_seqlen bounds the grid.
 * ---------------------------------------------------------------------------------------------- */
int omni_flash_attn_fwd(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out,
                        int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* cu_seqlens,
                        int32_t B, int32_t H, int32_t head_dim, int32_t max_seqlen, float softmax_scale,
                        omni_stream stream);
/* Same, with the output optionally K32-blocked: out_k32_rows = R > 0 -> out is [H*128/32][R][32] over the R rows of the
 * whole buffer (ldo ignored); 0 -> row-major. */
int omni_flash_attn_fwd_ex(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out,
                        int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* cu_seqlens,
                        int32_t B, int32_t H, int32_t head_dim, int32_t max_seqlen, float softmax_scale,
                        int32_t out_k32_rows, omni_stream stream);

/* ABI v11 — the same with an optional caller-owned fp32 DEVICE workspace (16-byte aligned, omni_flash_attn_workspace_bytes(B, H)
 * bytes; NULL = omni_flash_attn_fwd_ex).  With it, a grid whose per-item LAST 256-query block is short (<= 64 rows: 4160 = 16 x 256 +
 * 64 joint rows at 1024^2, 16448 = 64 x 256 + 64 at 2048^2) and would open a thin extra round of workgroups (one 2048^2 request:
 * 48 x 65 = 3120 workgroups = 12 rounds of 256 CUs + 48) splits that block's KEY range over s workgroups (s <= 8, picked from a
 * makespan model), which leave un-normalised fp32 partials (O, running max, row sum) in the workspace; a second small kernel merges
 * them (flash-decoding style).  Results agree with the unsplit kernel to fp32 summation order on those rows, bit for bit elsewhere. */
size_t omni_flash_attn_workspace_bytes(int32_t B, int32_t H);
int omni_flash_attn_fwd_ws(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out,
                           int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* cu_seqlens,
                           int32_t B, int32_t H, int32_t head_dim, int32_t max_seqlen, float softmax_scale,
                           int32_t out_k32_rows, void* workspace, size_t workspace_bytes, omni_stream stream);

/* ABI v11 — the attention of the WHOLE plug-in point: everything `SDPAImpl.forward` hands to
 * F.scaled_dot_product_attention (vllm_omni/diffusion/attention/backends/sdpa.py:46-66; the backend selector is process-global,
 * attention/selector.py:49-77, so a registered backend also serves the other in-tree DiTs) beyond the self-attention above:
 * separate query / key sequences (cross-attention, wan2_2_transformer.py:243,340), head size 64 or 128 (sd3_transformer.py:108),
 * an attention mask, is_causal, grouped K / V heads.
 *   q, out: [total_q_rows, H * dh]; k, v: [total_k_rows, H_kv * dh]; item b owns query rows [cu_seqlens_q[b], cu_seqlens_q[b+1])
 *   and key rows [cu_seqlens_k[b], cu_seqlens_k[b+1]) (DEVICE int32 arrays of B + 1 entries; the two may be the same array).
 *   mask (nullable, mask_type 0): element (b, h, i, j) at mask[b * mask_stride_b + h * mask_stride_h + i * mask_stride_q +
 *   j * mask_stride_k] with strides in ELEMENTS, 0 = broadcast (what `mask.expand(B, H, S_q, S_k).stride()` returns);
 *   mask_type 1: one byte per element, non-zero = attend (torch.bool); 2: bf16 added to the scaled scores; 3: fp32 added.
 *   causal = 1: key j takes part in query i only if j <= i (torch's is_causal: aligned to the top-left corner).
 *   out = softmax(scale * q k^T + mask) v per (item, head), fp32 softmax; rows with no key to attend to produce zeros.
 * head_dim 64 or 128 (OMNI_ERR_UNSUPPORTED otherwise); H % H_kv == 0. */
typedef struct {
  const omni_bf16 *q, *k, *v;
  omni_bf16* out;
  int64_t ldq, ldk, ldv, ldo;          /* row strides in elements (multiples of 8; ldo of 4) */
  const int32_t* cu_seqlens_q;
  const int32_t* cu_seqlens_k;
  int32_t B, H, H_kv, head_dim;
  int32_t max_seqlen_q, max_seqlen_k;  /* bounds of the per-item lengths (max_seqlen_q sizes the grid) */
  float softmax_scale;
  int32_t causal;
  const void* mask;
  int32_t mask_type;                   /* 0 none, 1 bool bytes, 2 bf16 additive, 3 fp32 additive */
  int64_t mask_stride_b, mask_stride_h, mask_stride_q, mask_stride_k;
} omni_attn_params;
int omni_flash_attn_general(const omni_attn_params* p, omni_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Small-batch weight-streaming linear:  y[b, n] = act_out( sum_k act_in(x[b, k]) * W[n, k] + bias[n] ).
 * Replaces the GEMVs at qwen_image_transformer.py:51-52 (TimestepEmbedding), :552,557
 * (img_mod / txt_mod = SiLU -> Linear(D -> 6D), 13.6 GB of weights per forward) and the Linear inside
 * AdaLayerNormContinuous (:797).  HBM-bound: W is read exactly once.  B <= 8, K % 8 == 0, K <= 4096.
 * act codes: 0 none, 1 SiLU.
 * ---------------------------------------------------------------------------------------------- */
int omni_linear_smallbatch(const omni_bf16* x, int64_t ldx, int32_t B, const omni_bf16* W, const omni_bf16* bias,
                           int64_t N, int32_t K, omni_bf16* y, int64_t ldy, int32_t act_in, int32_t act_out,
                           omni_stream stream);

/* Sinusoidal timestep projection, diffusers Timesteps(256, flip_sin_to_cos=True, shift 0, scale 1000)
 * (qwen_image_transformer.py:44,51; body in-tree at pipeline_qwen_image.py:135-184): out[b] = [cos | sin]. */
int omni_timestep_sinusoid(const float* t, int32_t B, int32_t dim, float scale, omni_bf16* out, omni_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Step epilogue: true-CFG combine + norm rescale + Flow-Match Euler update, fused.
 * Replaces pipeline_qwen_image.py:580-583 and scheduler.step at :585:
 *   comb = neg + s*(pos - neg); pred = comb * (||pos|| / ||comb||)   (norms over the last dim C)
 *   latents <- bf16( float(latents) + dt * pred )            (neg == NULL: pred = pos)
 * pos/neg/latents: [rows, C] bf16 contiguous, C == 64.  dt = sigma_next - sigma (per-row item via dt_item
 * when dt_rows_per_item > 0: dt[row / dt_rows_per_item], else dt[0]).  dt is a DEVICE fp32 array.
 * ---------------------------------------------------------------------------------------------- */
int omni_cfg_euler_step(const omni_bf16* pos, const omni_bf16* neg, omni_bf16* latents, int32_t rows, int32_t C,
                        float true_cfg_scale, const float* dt, int32_t dt_rows_per_item, omni_stream stream);
/* ABI v9 — same with the norm rescale optional: normalize = 0 -> pred = comb (the Layered pipeline's default,
 * pipeline_qwen_image_layered.py:599-605 `cfg_normalize`); normalize = 1 == omni_cfg_euler_step. */
int omni_cfg_euler_step_ex(const omni_bf16* pos, const omni_bf16* neg, omni_bf16* latents, int32_t rows, int32_t C,
                           float true_cfg_scale, const float* dt, int32_t dt_rows_per_item, int32_t normalize,
                           omni_stream stream);

/* ------------------------------------------------------------------------------------------------
 * VAE kernels (AutoencoderKLQwenImage.decode — and .encode for the Edit pipelines — for one frame,
 * vllm_omni/diffusion/models/qwen_image/autoencoder_kl_qwenimage.py:839-863 / diffusers twin).
 * Activations are NHWC bf16 ([B, H, W, C]); conv weights are pre-packed [Cout, 3, 3, Cin] bf16 from temporal
 * slice [-1] of the causal Conv3d weights (:69-84: the two zero front frames make the other slices dead).
 * ---------------------------------------------------------------------------------------------- */
/* 3x3 (pad 1) or 1x1 convolution as implicit GEMM on MFMA.  Optional fused prologue
 * RMS-norm(channels)*gamma -> SiLU on the input (QwenImageRMS_norm :108-109 + SiLU, :262-265),
 * optional nearest-exact x2 upsample of the input (QwenImageUpsample :112-124), optional residual add. */
typedef struct {
  const omni_bf16* x;     /* [B, Hin, Win, Cin] */
  const omni_bf16* w;     /* [Cout, ks, ks, Cin] */
  const omni_bf16* bias;  /* [Cout], nullable */
  const omni_bf16* gamma; /* [Cin], nullable -> no norm/SiLU prologue */
  const omni_bf16* res;   /* [B, Hout, Wout, Cout], nullable */
  omni_bf16* y;           /* [B, Hout, Wout, Cout] */
  int32_t B, Hin, Win, Cin, Cout;
  int32_t ksize;          /* 1 or 3 */
  int32_t upsample2x;     /* 1: Hout = 2*Hin (input index = out/2), else Hout = Hin */
  int32_t silu;           /* with gamma: apply SiLU after the norm */
  float clamp_lo, clamp_hi; /* applied if clamp_lo < clamp_hi */
  int32_t downsample2x;   /* 1 (3x3 only): ZeroPad2d((0,1,0,1)) + stride-2 conv of the VAE ENCODER's downsample2d/3d
                           * (autoencoder_kl_qwenimage.py:162-166): Hout = Hin / 2, source = 2*dst + tap.  ABI v3 */
  int32_t x_padded;       /* ABI v8.  1: x (and res) are ZERO-BORDERED rasters [B, Hin + 2, Win + 2, C]: the conv's zero padding
                           * (F.pad in QwenImageCausalConv3d.forward, :80-84) is resident in memory instead of re-created per tap */
  int32_t y_padded;       /* ABI v8.  1: y is written as a zero-bordered raster too (needs x_padded, Cin % 32 == 0,
                           * Cout % 8 == 0, no downsample): the decoder's 3x3 / 1x1 convs run as a GEMM over nine row-shifted
                           * views of x, LDS-DMA fed (vae.hip conv_bordered_kernel).  ABI v10: with upsample2x (3x3, no res) x is
                           * the half-resolution bordered raster [B, Hin + 2, Win + 2, Cin] and y [B, 2 Hin + 2, 2 Win + 2, Cout]:
                           * the operand fetch reads source pixel ((Y - 1) / 2 + 1, (X - 1) / 2 + 1) for pixel (Y, X) of the
                           * upsampled raster — the same result as omni_vae_upsample2x_bordered followed by the conv, bit for bit */
  /* ABI v10: the norm + activation that FOLLOWS this conv in the decoder (QwenImageResidualBlock.forward norm1 / norm2 + SiLU,
   * autoencoder_kl_qwenimage.py:262-275; norm_out :735-737) as a second output of the same launch:
   *   y_norm = silu?(F.normalize(y, dim=C) * sqrt(C) * norm_gamma)      computed from y as rounded to bf16
   * (== omni_vae_rmsnorm_silu(y) bit for bit up to the summation order of the squares).  norm_gamma NULL -> off. */
  const omni_bf16* norm_gamma; /* [Cout], nullable */
  omni_bf16* y_norm;           /* same shape as y; required with norm_gamma */
  int32_t norm_silu;           /* apply SiLU after the norm */
} omni_conv_params;
/* With norm_gamma: the conv kernel itself writes y_norm when one workgroup holds all channels of a pixel
 * (omni_vae_conv2d_fuses_norm(p) == 1: zero-bordered rasters, Cout <= 96, or Cout <= 192 on rasters that fill the chip with
 * 192-channel tiles); then y may be NULL (only the normed output is wanted).  Otherwise the call runs the conv into y (required)
 * and omni_vae_rmsnorm_silu from y into y_norm. */
int omni_vae_conv2d(const omni_conv_params* p, omni_stream stream);
int omni_vae_conv2d_fuses_norm(const omni_conv_params* p);

/* Nearest-exact x2 upsample (QwenImageUpsample, autoencoder_kl_qwenimage.py:112-124) between zero-bordered rasters:
 * x [B, H + 2, W + 2, C] -> y [B, 2H + 2, 2W + 2, C].  C % 8 == 0.  ABI v8 */
int omni_vae_upsample2x_bordered(const omni_bf16* x, omni_bf16* y, int32_t B, int32_t H, int32_t W, int32_t C,
                                 omni_stream stream);

/* Channel RMS-norm (F.normalize(dim=C) * sqrt(C) * gamma) with optional SiLU: NHWC rows of C channels. */
int omni_vae_rmsnorm_silu(const omni_bf16* x, omni_bf16* y, int64_t rows, int32_t C, const omni_bf16* gamma,
                          int32_t silu, omni_stream stream);

/* ABI v10 — the single-head attention of the VAE mid block (QwenImageAttentionBlock.forward,
 * autoencoder_kl_qwenimage.py:305-330: F.scaled_dot_product_attention over all tokens of an image, ONE head of C = 384 channels)
 * as one flash kernel: q, k, v [B][tokens][C] bf16 with row strides ldq / ldk / ldv (elements; images tokens * ld apart, so the
 * three may be column slices of one fused [B * tokens, 3 C] projection), out [B][tokens][C] with row stride ldo;
 * out = softmax(scale * q k^T) v per image.  C must be 384 (OMNI_ERR_UNSUPPORTED otherwise: callers fall back to
 * GEMM -> omni_softmax_rows -> GEMM); any token count (ragged tails are masked); tokens * ld * 2 < 4 GiB. */
int omni_vae_attention(const omni_bf16* q, const omni_bf16* k, const omni_bf16* v, omni_bf16* out, int32_t B, int32_t tokens,
                       int32_t C, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, omni_stream stream);

/* In-place row softmax p = softmax(scale * s) over rows of `cols` bf16 scores (row stride ld).  Used for the
 * single-head mid-block attention of the VAE (F.scaled_dot_product_attention at
 * autoencoder_kl_qwenimage.py:319), which runs as GEMM(QK^T) -> this -> GEMM(PV).  cols % 8 == 0. */
int omni_softmax_rows(omni_bf16* s, int64_t ld, int64_t rows, int32_t cols, float scale, omni_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Whole DiT forward as one native call: enqueues every kernel of
 * QwenImageTransformer2DModel.forward (qwen_image_transformer.py:692-802) on `stream` from a
 * descriptor of device pointers.  No host synchronisation, no allocation; hipGraph-capturable.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const omni_bf16 *img_mod_w, *img_mod_b, *txt_mod_w, *txt_mod_b;   /* [6D, D], [6D] */
  const omni_bf16 *to_qkv_w, *to_qkv_b, *add_qkv_w, *add_qkv_b;     /* [3D, D], [3D] */
  const omni_bf16 *norm_q_w, *norm_k_w, *norm_added_q_w, *norm_added_k_w; /* [128] */
  const omni_bf16 *to_out_w, *to_out_b, *to_add_out_w, *to_add_out_b; /* [D, D], [D] */
  const omni_bf16 *img_mlp_w1, *img_mlp_b1, *img_mlp_w2, *img_mlp_b2; /* [4D, D],[4D],[D,4D],[D] */
  const omni_bf16 *txt_mlp_w1, *txt_mlp_b1, *txt_mlp_w2, *txt_mlp_b2;
} omni_dit_layer_weights;

/* ABI v7 — optional fp8 copies of a layer's eight GEMM weights (BASELINE.json config 5): OCP e4m3 bytes in the K64-blocked
 * order + one fp32 scale per output channel, both written by omni_quantize_fp8_rows(weight [out, in]).  With
 * omni_dit_weights.fp8_layers set, every block GEMM runs as omni_gemm_params.fp8: its bf16 input (AdaLN output, attention
 * output, GELU output) is quantised per token by omni_quantize_fp8_rows right in front of it; biases, epilogues, the residual
 * streams, q/k/v, the attention and everything outside the block GEMMs stay bf16.
 * ABI v9 — per GEMM class: the four classes are (to_qkv, add_qkv), (to_out, to_add_out), (img/txt mlp w1), (img/txt mlp w2);
 * a class whose two weight pointers are NULL runs in bf16 from omni_dit_layer_weights as without fp8 (mixed recipes). */
typedef struct {
  const uint8_t *to_qkv_w8, *add_qkv_w8, *to_out_w8, *to_add_out_w8, *img_mlp_w1_8, *img_mlp_w2_8, *txt_mlp_w1_8, *txt_mlp_w2_8;
  const float *to_qkv_s, *add_qkv_s, *to_out_s, *to_add_out_s, *img_mlp_w1_s, *img_mlp_w2_s, *txt_mlp_w1_s, *txt_mlp_w2_s;
} omni_dit_fp8_layer;

typedef struct {
  int32_t num_layers, num_heads, head_dim, joint_dim, in_channels, out_channels_packed; /* 60,24,128,3584,64,64 */
  int32_t gemm_w_k32_blocked; /* 1: the eight [out,in] matrices of every layer (to_qkv, add_qkv, to_out, to_add_out,
                               * img/txt mlp w1, w2) are stored K32-blocked (omni_gemm_params.w_k32_blocked); all other
                               * weights are always row-major.  ABI v2. */
  const omni_bf16 *t_lin1_w, *t_lin1_b, *t_lin2_w, *t_lin2_b; /* TimestepEmbedding */
  const omni_bf16 *txt_norm_w, *img_in_w, *img_in_b, *txt_in_w, *txt_in_b;
  const omni_bf16 *norm_out_w, *norm_out_b, *proj_out_w, *proj_out_b;
  const omni_dit_layer_weights* layers; /* HOST array [num_layers] of device pointers */
  const omni_dit_fp8_layer* fp8_layers; /* ABI v7: nullable HOST array [num_layers]; NULL = bf16 GEMMs */
} omni_dit_weights;

/* ------------------------------------------------------------------------------------------------
 * TeaCache state of one step-batch (vllm_omni/diffusion/cache/teacache/hook.py:82-217, state.py, config.py), kept ON THE
 * DEVICE and evaluated by kernels inside omni_dit_forward, PER ITEM (request x CFG branch — the reference keeps one state
 * per CFG branch of its single request, hook.py:113-121):
 *   rel = mean|mod - prev_mod| / (mean|prev_mod| + 1e-8)   over the item's first-block modulated input (:195-206)
 *   acc += |poly(rel)|;  cnt == 0 -> compute (acc = 0);  acc < thresh -> SKIP the 60 blocks, else compute and acc = 0
 *   skip   : hidden_img += prev_res                       (:127-134; the encoder residual is dead code there: only
 *   compute: prev_res = hidden_img_after - hidden_img_before   (:135-157)   hidden_states reaches norm_out / proj_out)
 * The reference reads the decision back with .cpu().item() (a host sync per forward); here the decision stays on the
 * device: skipped items' GEMM row tiles and attention blocks return immediately (omni_gemm_group.tile_skip), so the launch
 * sequence is fixed, sync-free and hipGraph-capturable, and step-batched items keep their own B=1 decisions.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  float rel_l1_thresh;
  float coeff[5];        /* polynomial, highest power first (numpy.poly1d order, config.py:9-31) */
  omni_bf16* prev_mod;   /* [n_img_rows * D] previous modulated input (layout = whatever the first AdaLN writes) */
  omni_bf16* prev_res;   /* [n_img_rows, D] cached residual of the image stream */
  float* acc_dist;       /* [n_items] accumulated rescaled distance */
  int32_t* cnt;          /* [n_items] forwards seen since reset (0 -> always compute) */
  int32_t* skip;         /* [n_items] out: this forward's decision (1 = blocks skipped) */
  int32_t* skip_total;   /* [n_items] out: running count of skipped forwards (statistics) */
  float* scratch;        /* [2 * n_items] zero-initialised partial sums (left zeroed by every forward) */
  int32_t* tile_skip_img;/* [ceil(n_img_rows / 256)] scratch */
  int32_t* tile_skip_txt;/* [ceil(n_txt_rows / 256)] scratch */
  const int32_t* txt_cu; /* [n_items + 1] prefix sums of the text rows */
} omni_teacache;

typedef struct {
  /* ragged batch of n_items sequences; item i: T_i text rows then S_img image rows in the joint order */
  int32_t n_items, n_img_rows, n_txt_rows, n_joint_rows, n_temb, max_seqlen;
  const omni_bf16* latents;       /* [n_img_rows, in_channels] packed latents           */
  const omni_bf16* prompt_embeds; /* [n_txt_rows, joint_dim]                            */
  const float* timestep;          /* [n_temb] sigma (the pipeline's t/1000)             */
  const int32_t* cu_seqlens;      /* [n_items+1] joint-row prefix sums                  */
  const int32_t* img_item;        /* [n_img_rows] -> temb row of that token             */
  const int32_t* txt_item;        /* [n_txt_rows] -> temb row                            */
  const int32_t* img_joint_row;   /* [n_img_rows] -> joint row                           */
  const int32_t* txt_joint_row;   /* [n_txt_rows] -> joint row                           */
  const int32_t* joint_pos;       /* [n_joint_rows] -> rope table row                    */
  int32_t txt_pos_end;            /* rope rows < this are text positions                 */
  const omni_bf16 *rope_cos, *rope_sin; /* [npos, 64] bf16                               */
  omni_bf16* noise_pred;          /* [n_img_rows, out_channels_packed]                   */
  /* workspace (caller-allocated, sizes from omni_dit_workspace_bytes) */
  void* workspace;
  size_t workspace_bytes;
  const omni_teacache* teacache;  /* nullable: TeaCache off.  All items must have the same number of image rows.  ABI v3 */
  const omni_bf16* temb_add;      /* ABI v9, nullable: [n_temb, D] bf16 added to the timestep embedding — the Layered variant's
                                   * addition_t_embedding(additional_t_cond) rows (qwen_image_transformer.py:47-62) */
  const omni_bf16* mod_table;     /* ABI v9, nullable: [num_layers][2 (image, text stream)][n_temb][6 D] bf16 — the blocks'
                                   * modulation vectors for THIS forward's conditioning rows, taken from
                                   * omni_dit_modulation_table; the 2 x num_layers weight-streaming GEMVs are then skipped */
} omni_dit_batch;

size_t omni_dit_workspace_bytes(const omni_dit_weights* w, int32_t n_img_rows, int32_t n_txt_rows, int32_t n_temb);

/* ABI v9 — the modulation vectors of every block for M conditioning rows in ONE pass over the modulation weights:
 *   table[l][s][m] = Linear_{l,s}(SiLU(temb[m]))   s = 0: img_mod, 1: txt_mod   (qwen_image_transformer.py:478-481,494-497,552-561)
 * temb [M, D] bf16 = the timestep embeddings (time_text_embed) of all the denoising steps of a request — they are known
 * before the loop starts — table [num_layers][2][M][6 D] bf16.  The reference re-streams these 13.6 GB of weights in EVERY
 * forward (226 MB per block for 0.23 GFLOP); a request of N steps x 2 CFG branches reads them 2N times, here
 * once.  At 256^2 (a forward is ~10 ms of which 2.7 ms is this stream) that is a quarter of the latency.  SiLU output is
 * rounded to bf16 before the product, as in the reference's eager nn.Sequential(SiLU, Linear).  Caller-owned workspace. */
size_t omni_dit_modulation_table_workspace_bytes(const omni_dit_weights* w, int32_t M);
int omni_dit_modulation_table(const omni_dit_weights* w, const omni_bf16* temb, int32_t M, omni_bf16* table, void* workspace,
                              size_t workspace_bytes, omni_stream stream);
int omni_dit_forward(const omni_dit_weights* w, const omni_dit_batch* b, omni_stream stream);

/* One dual-stream block (QwenImageTransformerBlock.forward, qwen_image_transformer.py:541-605) on caller-owned residual
 * streams: hidden_img [n_img_rows, D], hidden_txt [n_txt_rows, D] (updated in place), temb [n_temb, D].  Same kernels and
 * the same batch descriptor as omni_dit_forward (latents / prompt_embeds / timestep / noise_pred of `b` are not read).
 * This is the entry point behind the module-level plug-in surface that cache hooks walk
 * (cache/teacache/extractors.py:216-233 calls `block(hidden_states=..., encoder_hidden_states=..., temb=...)`). */
int omni_dit_block(const omni_dit_weights* w, int32_t layer, const omni_dit_batch* b, omni_bf16* hidden_img,
                   omni_bf16* hidden_txt, const omni_bf16* temb, omni_stream stream);

/* The two halves of a block around the attention, for SEQUENCE-PARALLEL callers (Ulysses: all-to-all of q/k/v before the
 * attention and of its output after it — reference vllm_omni/diffusion/attention/parallel/ulysses.py:59-135,
 * qwen_image_transformer.py:735-742,776-781,800-801).  `b` describes THIS RANK's rows (its chunk of the image tokens + the
 * replicated text tokens).  omni_dit_block_qkv runs modulation + norm1 + the fused QKV projection (+ q/k norm + RoPE) and
 * returns pointers to the joint-order [n_joint_rows, D] q/k/v inside the workspace; the caller exchanges them, runs
 * omni_flash_attn_fwd on the head slice it owns, exchanges back, and hands the row-major [n_joint_rows, D] attention output of
 * its own rows to omni_dit_block_post, which finishes the block (output projections, gated residuals, norm2, MLP). */
int omni_dit_block_qkv(const omni_dit_weights* w, int32_t layer, const omni_dit_batch* b, omni_bf16* hidden_img,
                       omni_bf16* hidden_txt, const omni_bf16* temb, omni_bf16** q, omni_bf16** k, omni_bf16** v,
                       omni_stream stream);
int omni_dit_block_post(const omni_dit_weights* w, int32_t layer, const omni_dit_batch* b, omni_bf16* hidden_img,
                        omni_bf16* hidden_txt, const omni_bf16* temb, const omni_bf16* attn, omni_stream stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* OMNI_CDNA4_H */
