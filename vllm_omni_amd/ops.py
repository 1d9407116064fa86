"""Tensor-level wrappers over the C-ABI (raw device pointers + the current HIP stream).

PyTorch is plumbing here (device memory, streams); all arithmetic happens in libomni_cdna4.so.
Every wrapper raises if a tensor is not a bf16/int32/fp32 CUDA(HIP) tensor of the expected layout —
there is no eager fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N

BF16 = torch.bfloat16

EPI_BIAS, EPI_BIAS_GELU_TANH, EPI_BIAS_GATE_RES, EPI_BIAS_SPLIT3, EPI_BIAS_SPLIT3_QKNORM_ROPE = 0, 1, 2, 3, 4


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: torch.Tensor | None, dtype=BF16, name: str = "tensor") -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        raise N.OmniNativeError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise N.OmniNativeError(f"{name} must be {dtype}, got {t.dtype}")
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise N.OmniNativeError(f"{name} must be contiguous in its last dimension")
    return t.data_ptr()


def _rows2d(t: torch.Tensor, name: str) -> tuple[int, int, int]:
    """(rows, cols, row_stride) of a 2-D row-major view."""
    if t.dim() != 2:
        raise N.OmniNativeError(f"{name} must be 2-D, got shape {tuple(t.shape)}")
    return t.shape[0], t.shape[1], t.stride(0)


class GemmGroupArgs:
    """Python-side mirror of omni_gemm_group (one stream of a grouped GEMM)."""

    def __init__(self, a, w, bias=None, out=None, *, a_row_map=None, out_row_map=None, out1=None, out2=None,
                 res=None, gate=None, gate_item_stride=0, row_item_map=None, rows_per_item=0,
                 a_k32_blocked=False, out_k32_blocked=False, qk_norm_q_w=None, qk_norm_k_w=None, qk_rope_cos=None,
                 qk_rope_sin=None, qk_row_pos=None, qk_eps=1e-6, a_scale=None, w_scale=None, tile_skip=None):
        self.a_k32_blocked, self.out_k32_blocked = a_k32_blocked, out_k32_blocked
        self.a_scale, self.w_scale = a_scale, w_scale            # fp8 operands (gemm(..., fp8=True)): fp32 row / channel scales
        self.qk = (qk_norm_q_w, qk_norm_k_w, qk_rope_cos, qk_rope_sin, qk_row_pos, qk_eps)
        self.a, self.w, self.bias, self.out = a, w, bias, out
        self.a_row_map, self.out_row_map, self.out1, self.out2 = a_row_map, out_row_map, out1, out2
        self.res, self.gate, self.gate_item_stride = res, gate, gate_item_stride
        self.row_item_map, self.rows_per_item = row_item_map, rows_per_item
        self.tile_skip = tile_skip                                # int32 [ceil(M / 256)] device flags: non-zero = that row tile is skipped


GEMM_KERNEL_AUTO, GEMM_KERNEL_RING, GEMM_KERNEL_SPLITK_TALL, GEMM_KERNEL_NO_TAIL_SPLIT = 0, 1, 2, 3      # omni_gemm_params.kernel_hint
GEMM_KERNEL_SPLITK_IN_LAUNCH = 4                                                                      # ABI v12
GEMM_KERNEL_SPLITK_DEFER_FINISH = 5                                                                   # ABI v13


def w_to_k32_blocked(w: torch.Tensor) -> torch.Tensor:
    """[N, K] row-major -> the same values in K32-blocked order [K/32][N][32] (omni_gemm_params.w_k32_blocked = 1),
    returned as an [N, K]-shaped contiguous tensor so that shape checks and pointer plumbing stay unchanged."""
    n, k = w.shape
    if k % 32:
        raise ValueError("K must be a multiple of 32")
    return w.view(n, k // 32, 32).transpose(0, 1).contiguous().view(n, k)


def gemm(groups: list[GemmGroupArgs], epilogue: int = EPI_BIAS, split_n: int = 0, m_override: list[int] | None = None,
         w_k32_blocked: bool = False, splitk_ws: torch.Tensor | None = None, kernel_hint: int = 0, fp8: bool = False):
    p = _gemm_params(groups, epilogue, split_n, m_override, w_k32_blocked, splitk_ws, kernel_hint, fp8)
    N.check(N.lib().omni_gemm_bf16(C.byref(p), _stream()), "omni_gemm_bf16")


def gemm_splitk_factor(groups: list[GemmGroupArgs], epilogue: int = EPI_BIAS, split_n: int = 0, m_override: list[int] | None = None,
                       w_k32_blocked: bool = False, splitk_ws: torch.Tensor | None = None, kernel_hint: int = 0,
                       fp8: bool = False) -> int:
    """omni_gemm_splitk_factor (ABI v13): the whole-launch split-K factor `gemm` would use for these arguments (1 = no split).
    A caller that wants the partials left for `splitk_finish_adaln_pair` asks this first and then calls `gemm` with
    kernel_hint = GEMM_KERNEL_SPLITK_DEFER_FINISH."""
    p = _gemm_params(groups, epilogue, split_n, m_override, w_k32_blocked, splitk_ws, kernel_hint, fp8)
    s = int(N.lib().omni_gemm_splitk_factor(C.byref(p)))
    if s <= 0:
        raise N.OmniNativeError("omni_gemm_splitk_factor: not a valid GEMM call")
    return s


def _gemm_params(groups, epilogue, split_n, m_override, w_k32_blocked, splitk_ws, kernel_hint, fp8):
    """Y_g = epilogue(A_g @ W_g.T + bias_g) for up to two groups sharing N, K (omni_gemm_bf16).  `splitk_ws`: optional fp32
    device workspace; with it, launches of at most 128 tiles in at most 10 row tiles split their K loop (ABI v4).
    `kernel_hint`: 0 = automatic, GEMM_KERNEL_RING = force the fallback (ring) kernel (ABI v6; cross-checks),
    GEMM_KERNEL_SPLITK_TALL = automatic + split-K also for tall launches (ABI v10).
    `fp8`: A and W are uint8 tensors of OCP e4m3 values in the K64-blocked order (`quantize_fp8_rows`), with fp32
    `a_scale` / `w_scale` per group (ABI v7): Y = epilogue((A8 @ W8.T) * a_scale[:, None] * w_scale[None, :] + bias)."""
    p = N.GemmParams()
    p.kernel_hint = int(kernel_hint)
    p.fp8 = 1 if fp8 else 0
    op_dtype = torch.uint8 if fp8 else BF16
    if splitk_ws is not None:
        if splitk_ws.dtype != torch.float32 or not splitk_ws.is_cuda or not splitk_ws.is_contiguous():
            raise N.OmniNativeError("splitk_ws must be a contiguous float32 GPU tensor")
        p.splitk_ws, p.splitk_ws_floats = splitk_ws.data_ptr(), splitk_ws.numel()
    p.ngroups = len(groups)
    p.epilogue = epilogue
    p.split_n = split_n
    p.w_k32_blocked = 1 if w_k32_blocked else 0
    N_, K_ = groups[0].w.shape
    p.N, p.K = N_, K_
    for i, g in enumerate(groups):
        G = p.g[i]
        m, k, lda = _rows2d(g.a, "A")
        if g.w.shape != (N_, K_) or not g.w.is_contiguous():
            raise N.OmniNativeError("all groups must share a contiguous [N, K] weight shape")
        G.A, G.lda = _p(g.a, op_dtype, name="A"), lda
        G.M = m_override[i] if m_override else (g.a_row_map.numel() if g.a_row_map is not None else m)
        if g.a_row_map is None and k != K_:
            raise N.OmniNativeError(f"A has K={k}, W has K={K_}")
        G.a_row_map = _p(g.a_row_map, torch.int32, "a_row_map")
        G.W, G.bias = _p(g.w, op_dtype, name="W"), _p(g.bias, name="bias")
        if fp8:
            G.a_scale, G.w_scale = _p(g.a_scale, torch.float32, "a_scale"), _p(g.w_scale, torch.float32, "w_scale")
        G.out, G.ldo = _p(g.out, name="out"), g.out.stride(0)
        G.out1, G.out2 = _p(g.out1, name="out1"), _p(g.out2, name="out2")
        G.out_row_map = _p(g.out_row_map, torch.int32, "out_row_map")
        if g.res is not None:
            G.res, G.ldres = _p(g.res, name="res"), g.res.stride(0)
        G.gate, G.gate_item_stride = _p(g.gate, name="gate"), g.gate_item_stride
        G.row_item_map = _p(g.row_item_map, torch.int32, "row_item_map")
        G.rows_per_item = g.rows_per_item
        qw, kw, cos, sin, pos, eps = g.qk
        G.qk_norm_q_w, G.qk_norm_k_w = _p(qw, name="qk_norm_q_w"), _p(kw, name="qk_norm_k_w")
        G.qk_rope_cos, G.qk_rope_sin = _p(cos, name="qk_rope_cos"), _p(sin, name="qk_rope_sin")
        G.qk_row_pos, G.qk_eps = _p(pos, torch.int32, "qk_row_pos"), eps
        G.a_k32_rows = g.a.shape[0] if g.a_k32_blocked else 0       # blocked tensors keep their [rows, K] shape
        G.out_k32_rows = g.out.shape[0] if g.out_k32_blocked else 0
        G.tile_skip = _p(g.tile_skip, torch.int32, "tile_skip")
    return p


def quantize_fp8_rows(x: torch.Tensor, *, x_k32_blocked: bool = False, out: torch.Tensor | None = None,
                      scale: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """Dynamic per-row fp8 quantisation (omni_quantize_fp8_rows): x [rows, K] bf16 (row-major, or K32-blocked with
    x_k32_blocked) -> (y8 uint8 [rows, K] holding e4m3 bytes in the K64-blocked order [K/64][rows][64], scale fp32 [rows]) with
    x[r, k] ~= e4m3(y8)[r, k] * scale[r].  Rows of activations = tokens; rows of a weight = output channels."""
    rows, K, ldx = _rows2d(x, "x")
    y8 = torch.empty(rows, K, dtype=torch.uint8, device=x.device) if out is None else out
    sc = torch.empty(rows, dtype=torch.float32, device=x.device) if scale is None else scale
    N.check(N.lib().omni_quantize_fp8_rows(_p(x, name="x"), ldx, rows if x_k32_blocked else 0, rows, K,
                                           _p(y8, torch.uint8, "y8"), y8.shape[0], _p(sc, torch.float32, "scale"), _stream()),
            "omni_quantize_fp8_rows")
    return y8, sc


def adaln_modulate_fp8(x, scale, shift, *, mod_item_stride: int, row_item_map=None, rows_per_item: int = 0, eps: float = 1e-6,
                       want_bf16: bool = False):
    """AdaLN-modulate with the fp8 quantisation fused in (omni_adaln_modulate_fp8): -> (y8 uint8 [rows, D] K64-blocked,
    scale fp32 [rows], y bf16 [rows, D] K32-blocked or None)."""
    rows, D, ldx = _rows2d(x, "x")
    y8 = torch.empty(rows, D, dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    y = torch.empty(rows, D, dtype=BF16, device=x.device) if want_bf16 else None
    N.check(N.lib().omni_adaln_modulate_fp8(_p(x, name="x"), ldx, rows, D, _p(scale, name="scale"), _p(shift, name="shift"),
                                            mod_item_stride, _p(row_item_map, torch.int32, "row_item_map"), rows_per_item, eps,
                                            _p(y, name="y"), rows if want_bf16 else 0, _p(y8, torch.uint8, "y8"), rows,
                                            _p(sc, torch.float32, "scale_out"), _stream()), "omni_adaln_modulate_fp8")
    return y8, sc, y


def k64_blocked_fp8_to_rows(y8: torch.Tensor) -> torch.Tensor:
    """[rows, K]-shaped uint8 tensor holding [K/64][rows][64] -> row-major float8_e4m3fn view [rows, K] (tests / debugging)."""
    r, k = y8.shape
    return y8.view(k // 64, r, 64).transpose(0, 1).contiguous().view(r, k).view(torch.float8_e4m3fn)


def linear(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None, *, gelu: bool = False,
           w_k32_blocked: bool = False) -> torch.Tensor:
    """Single-group convenience: x [M,K] @ w[N,K].T + bias (optionally GELU-tanh)."""
    out = torch.empty(x.shape[0], w.shape[0], dtype=BF16, device=x.device)
    gemm([GemmGroupArgs(x, w, bias, out)], EPI_BIAS_GELU_TANH if gelu else EPI_BIAS, w_k32_blocked=w_k32_blocked)
    return out


def k32_blocked_to_rows(t: torch.Tensor) -> torch.Tensor:
    """Inverse of the K32-blocked activation / weight layout: an [R, K]-shaped tensor holding [K/32][R][32] -> row-major."""
    r, k = t.shape
    return t.view(k // 32, r, 32).transpose(0, 1).contiguous().view(r, k)


def adaln_modulate(x, scale, shift, *, mod_item_stride: int, row_item_map=None, rows_per_item: int = 0,
                   eps: float = 1e-6, out=None, out_k32_blocked: bool = False):
    """out_k32_blocked: y (same [rows, D] shape) holds the K32-blocked order [D/32][rows][32] (omni_adaln_modulate_ex)."""
    rows, D, ldx = _rows2d(x, "x")
    y = torch.empty(rows, D, dtype=BF16, device=x.device) if out is None else out
    N.check(N.lib().omni_adaln_modulate_ex(_p(x, name="x"), ldx, _p(y, name="y"), y.stride(0), rows, D,
                                           _p(scale, name="scale"), _p(shift, name="shift"), mod_item_stride,
                                           _p(row_item_map, torch.int32, "row_item_map"), rows_per_item, eps,
                                           y.shape[0] if out_k32_blocked else 0, _stream()),
            "omni_adaln_modulate_ex")
    return y


def adaln_modulate_pair(streams, *, mod_item_stride: int, eps: float = 1e-6, out_k32_blocked: bool = False, fp8: bool = False,
                        want_bf16: bool = True):
    """omni_adaln_modulate_pair (ABI v12): AdaLN-modulate of TWO row groups (the image and the text stream of a DiT block) in one
    launch.  `streams` = two (x [rows, D] contiguous, scale, shift, row_item_map) tuples.  Returns per stream what the single
    calls return: y (row-major, or K32-blocked with `out_k32_blocked`); with `fp8` (y8, scale, y-or-None: the bf16 copy is
    K32-blocked) as `adaln_modulate_fp8`."""
    assert len(streams) == 2
    recs, outs = (N.AdalnStream * 2)(), []
    D = streams[0][0].shape[1]
    for i, (x, scale, shift, item_map) in enumerate(streams):
        rows, d, ldx = _rows2d(x, "x")
        if d != D or ldx != D:
            raise N.OmniNativeError("adaln_modulate_pair: both streams are contiguous [rows, D] with one D")
        g = recs[i]
        g.x, g.rows, g.scale, g.shift = _p(x, name="x"), rows, _p(scale, name="scale"), _p(shift, name="shift")
        g.row_item_map, g.rows_per_item = _p(item_map, torch.int32, "row_item_map"), 0
        y = torch.empty(rows, D, dtype=BF16, device=x.device) if (want_bf16 or not fp8) else None
        g.y, g.y_k32_rows = _p(y, name="y"), rows if (y is not None and (out_k32_blocked or fp8)) else 0
        if fp8:
            y8 = torch.empty(rows, D, dtype=torch.uint8, device=x.device)
            sc = torch.empty(rows, dtype=torch.float32, device=x.device)
            g.y8, g.y8_rows, g.y8_scale = _p(y8, torch.uint8, "y8"), rows, _p(sc, torch.float32, "scale_out")
            outs.append((y8, sc, y))
        else:
            outs.append(y)
    N.check(N.lib().omni_adaln_modulate_pair(C.byref(recs[0]), C.byref(recs[1]), D, mod_item_stride, eps, _stream()),
            "omni_adaln_modulate_pair")
    return outs


def splitk_finish_adaln_pair(splitk_ws: torch.Tensor, nsplit: int, ws_rows: int, streams, *, mod_item_stride: int,
                             eps: float = 1e-6, out_k32_blocked: bool = False):
    """omni_splitk_finish_adaln_pair (ABI v13): the finish of a DEFERRED split-K GEMM with the gated-residual epilogue + the AdaLN
    that follows it, both streams of a DiT block in one launch.  `splitk_ws` = the fp32 partials [nsplit][ws_rows][D] the GEMM left;
    `streams` = two (ws_row0, bias-or-None, hidden [rows, D] contiguous, gate, scale, shift, row_item_map) tuples.  `hidden` is
    updated IN PLACE (hidden + gate * (sum of partials + bias)); returns the two AdaLN outputs (row-major, or K32-blocked)."""
    assert len(streams) == 2
    recs, outs = (N.FinishAdalnStream * 2)(), []
    D = streams[0][2].shape[1]
    for i, (row0, bias, hidden, gate, scale, shift, item_map) in enumerate(streams):
        rows, d, ld = _rows2d(hidden, "hidden")
        if d != D or ld != D:
            raise N.OmniNativeError("splitk_finish_adaln_pair: both streams are contiguous [rows, D] with one D")
        g = recs[i]
        g.rows, g.ws_row0, g.bias, g.hidden = rows, int(row0), _p(bias, name="bias"), _p(hidden, name="hidden")
        g.gate, g.scale, g.shift = _p(gate, name="gate"), _p(scale, name="scale"), _p(shift, name="shift")
        g.row_item_map, g.rows_per_item = _p(item_map, torch.int32, "row_item_map"), 0
        y = torch.empty(rows, D, dtype=BF16, device=hidden.device)
        g.y, g.y_k32_rows = _p(y, name="y"), rows if out_k32_blocked else 0
        outs.append(y)
    N.check(N.lib().omni_splitk_finish_adaln_pair(_p(splitk_ws, torch.float32, "splitk_ws"), int(nsplit), int(ws_rows),
                                                  C.byref(recs[0]), C.byref(recs[1]), D, mod_item_stride, eps, _stream()),
            "omni_splitk_finish_adaln_pair")
    return outs


def rmsnorm(x, weight, eps: float = 1e-6, out=None):
    rows, D, ldx = _rows2d(x, "x")
    y = torch.empty_like(x) if out is None else out
    N.check(N.lib().omni_rmsnorm(_p(x, name="x"), ldx, _p(y, name="y"), y.stride(0), rows, D, _p(weight, name="w"),
                                 eps, _stream()), "omni_rmsnorm")
    return y


def qk_norm_rope_(x, num_heads, w_img, w_txt, cos_tab, sin_tab, row_pos, txt_pos_end, eps: float = 1e-6):
    rows, _, ldx = _rows2d(x, "x")
    N.check(N.lib().omni_qk_norm_rope(_p(x, name="x"), ldx, rows, num_heads, _p(w_img, name="w_img"),
                                      _p(w_txt, name="w_txt"), _p(cos_tab, name="cos"), _p(sin_tab, name="sin"),
                                      _p(row_pos, torch.int32, "row_pos"), txt_pos_end, eps, _stream()),
            "omni_qk_norm_rope")
    return x


def rope_interleaved(x, cos_tab, sin_tab, out=None):
    """x [B,S,H,dh] contiguous bf16; cos/sin [S, dh/2] bf16."""
    if x.dim() != 4 or not x.is_contiguous():
        raise N.OmniNativeError("rope_interleaved expects a contiguous [B,S,H,dh] tensor")
    B, S, H, dh = x.shape
    y = torch.empty_like(x) if out is None else out
    N.check(N.lib().omni_rope_interleaved(_p(x, name="x"), _p(y, name="y"), B, S, H, dh,
                                          _p(cos_tab.contiguous(), name="cos"), _p(sin_tab.contiguous(), name="sin"),
                                          _stream()), "omni_rope_interleaved")
    return y


def flash_attn_varlen(q, k, v, cu_seqlens, num_heads: int, max_seqlen: int, softmax_scale: float, out=None,
                      out_k32_blocked: bool = False, workspace: torch.Tensor | None = None):
    """q,k,v [rows, H*128] (row strides free), cu_seqlens int32 [B+1] on device.  out_k32_blocked: out (same shape)
    holds the K32-blocked order [H*128/32][rows][32] (omni_flash_attn_fwd_ex).  `workspace`: optional fp32 device tensor of
    `flash_attn_workspace_floats(B, H)` elements — lets the kernel split the key range of a short last q-block
    (omni_flash_attn_fwd_ws, ABI v11)."""
    rows, HD, ldq = _rows2d(q, "q")
    o = torch.empty(rows, HD, dtype=BF16, device=q.device) if out is None else out
    wsp, wsb = (None, 0) if workspace is None else (_p(workspace, torch.float32, "workspace"), workspace.numel() * 4)
    N.check(N.lib().omni_flash_attn_fwd_ws(_p(q, name="q"), _p(k, name="k"), _p(v, name="v"), _p(o, name="out"), ldq,
                                           k.stride(0), v.stride(0), o.stride(0),
                                           _p(cu_seqlens, torch.int32, "cu_seqlens"), cu_seqlens.numel() - 1, num_heads,
                                           HD // num_heads, max_seqlen, softmax_scale,
                                           o.shape[0] if out_k32_blocked else 0, wsp, wsb, _stream()),
            "omni_flash_attn_fwd_ws")
    return o


def flash_attn_workspace_floats(B: int, H: int) -> int:
    return int(N.lib().omni_flash_attn_workspace_bytes(B, H)) // 4


_MASK_TYPES = {torch.bool: 1, torch.uint8: 1, torch.bfloat16: 2, torch.float32: 3}


def flash_attn_general(q, k, v, cu_seqlens_q, cu_seqlens_k, num_heads: int, num_kv_heads: int, max_seqlen_q: int,
                       max_seqlen_k: int, softmax_scale: float, causal: bool = False, mask: torch.Tensor | None = None,
                       mask_strides: tuple[int, int, int, int] | None = None, out=None):
    """omni_flash_attn_general (ABI v11): q / out [rows_q, H * dh], k / v [rows_k, H_kv * dh], dh 64 or 128; per-item row ranges
    from the two int32 prefix-sum vectors.  `mask`: bool / uint8 (True = attend), bf16 or fp32 (additive) DEVICE tensor whose
    element (b, h, i, j) sits `mask_strides` (in elements, 0 = broadcast) apart — `mask.expand(B, H, Sq, Sk).stride()`."""
    rows, HD, ldq = _rows2d(q, "q")
    o = torch.empty(rows, HD, dtype=BF16, device=q.device) if out is None else out
    p = N.AttnParams()
    p.q, p.k, p.v, p.out = _p(q, name="q"), _p(k, name="k"), _p(v, name="v"), _p(o, name="out")
    p.ldq, p.ldk, p.ldv, p.ldo = ldq, k.stride(0), v.stride(0), o.stride(0)
    p.cu_seqlens_q, p.cu_seqlens_k = _p(cu_seqlens_q, torch.int32, "cu_seqlens_q"), _p(cu_seqlens_k, torch.int32, "cu_seqlens_k")
    p.B, p.H, p.H_kv, p.head_dim = cu_seqlens_q.numel() - 1, num_heads, num_kv_heads, HD // num_heads
    p.max_seqlen_q, p.max_seqlen_k = max_seqlen_q, max_seqlen_k
    p.softmax_scale, p.causal = softmax_scale, int(bool(causal))
    if mask is not None:
        if mask.dtype not in _MASK_TYPES:
            raise N.OmniNativeError(f"attention mask must be bool, uint8, bfloat16 or float32, got {mask.dtype}")
        if not mask.is_cuda:
            raise N.OmniNativeError("attention mask must live on the GPU")
        if mask_strides is None or len(mask_strides) != 4:
            raise N.OmniNativeError("mask_strides = the four element strides of the mask over (B, H, S_q, S_k)")
        p.mask, p.mask_type = mask.data_ptr(), _MASK_TYPES[mask.dtype]
        p.mask_stride_b, p.mask_stride_h, p.mask_stride_q, p.mask_stride_k = (int(x) for x in mask_strides)
    N.check(N.lib().omni_flash_attn_general(C.byref(p), _stream()), "omni_flash_attn_general")
    return o


def linear_smallbatch(x, w, bias=None, act_in: int = 0, act_out: int = 0, out=None):
    B, K, ldx = _rows2d(x, "x")
    Nn = w.shape[0]
    y = torch.empty(B, Nn, dtype=BF16, device=x.device) if out is None else out
    N.check(N.lib().omni_linear_smallbatch(_p(x, name="x"), ldx, B, _p(w, name="W"), _p(bias, name="bias"), Nn, K,
                                           _p(y, name="y"), y.stride(0), act_in, act_out, _stream()),
            "omni_linear_smallbatch")
    return y


def timestep_sinusoid(t: torch.Tensor, dim: int = 256, scale: float = 1000.0):
    out = torch.empty(t.numel(), dim, dtype=BF16, device=t.device)
    N.check(N.lib().omni_timestep_sinusoid(_p(t, torch.float32, "t"), t.numel(), dim, scale, _p(out), _stream()),
            "omni_timestep_sinusoid")
    return out


def cfg_euler_step_(latents, pos, neg, true_cfg_scale: float, dt: torch.Tensor, dt_rows_per_item: int = 0,
                    normalize: bool = True):
    """latents [rows,64] bf16 updated in place; dt fp32 device tensor.  normalize=False: the true-CFG combination without the
    norm rescale (the Layered pipeline's default)."""
    rows, Cc, _ = _rows2d(latents, "latents")
    if not (latents.is_contiguous() and pos.is_contiguous() and (neg is None or neg.is_contiguous())):
        raise N.OmniNativeError("cfg_euler_step_ needs contiguous [rows, 64] tensors")
    N.check(N.lib().omni_cfg_euler_step_ex(_p(pos, name="pos"), _p(neg, name="neg"), _p(latents, name="latents"), rows,
                                           Cc, true_cfg_scale, _p(dt, torch.float32, "dt"), dt_rows_per_item,
                                           1 if normalize else 0, _stream()),
            "omni_cfg_euler_step_ex")
    return latents


def vae_conv2d(x, w, bias=None, *, gamma=None, silu=True, res=None, upsample2x=False, downsample2x=False, clamp=None,
               out=None, x_bordered=False, y_bordered=False, norm_gamma=None, norm_silu=True, keep_raw=True):
    """NHWC bf16 conv (3x3 pad 1 or 1x1): x [B,H,W,Cin], w [Cout,ks,ks,Cin].  downsample2x: the encoder's zero-pad
    (right/bottom) + stride-2 3x3 conv.  x_bordered: x (and res) are zero-bordered rasters [B,H+2,W+2,C]; y_bordered: so is
    the output (the decoder's LDS-DMA-fed shifted-GEMM kernel; needs x_bordered, Cin % 32 == 0, Cout % 8 == 0).

    norm_gamma [Cout]: ALSO return silu?(rmsnorm(y) * norm_gamma) — the norm + activation that follows this conv in the decoder —
    from the same launch where the kernel holds all channels of a pixel (omni_vae_conv2d_fuses_norm), from a second pass
    otherwise.  Returns (y, y_norm); with keep_raw=False y is None when the kernel did not have to write it."""
    B, Hin, Win, Cin = x.shape
    if x_bordered:
        Hin, Win = Hin - 2, Win - 2
    Cout, ks = w.shape[0], w.shape[1]
    Hout, Wout = (2 * Hin, 2 * Win) if upsample2x else ((Hin // 2, Win // 2) if downsample2x else (Hin, Win))
    shape = (B, Hout + 2 * int(y_bordered), Wout + 2 * int(y_bordered), Cout)
    p = N.ConvParams()
    p.x, p.w, p.bias, p.gamma = _p(x.contiguous(), name="x"), _p(w, name="w"), _p(bias, name="bias"), _p(gamma)
    p.res = _p(res, name="res")
    p.B, p.Hin, p.Win, p.Cin, p.Cout, p.ksize = B, Hin, Win, Cin, Cout, ks
    p.upsample2x, p.silu, p.downsample2x = int(upsample2x), int(silu), int(downsample2x)
    p.x_padded, p.y_padded = int(x_bordered), int(y_bordered)
    p.clamp_lo, p.clamp_hi = clamp if clamp else (0.0, 0.0)
    yn = None
    if norm_gamma is not None:
        yn = torch.empty(shape, dtype=BF16, device=x.device)
        p.norm_gamma, p.y_norm, p.norm_silu = _p(norm_gamma, name="norm_gamma"), _p(yn), int(norm_silu)
        keep_raw = keep_raw or out is not None or not N.lib().omni_vae_conv2d_fuses_norm(C.byref(p))
    y = None
    if norm_gamma is None or keep_raw:
        y = torch.empty(shape, dtype=BF16, device=x.device) if out is None else out
    p.y = _p(y, name="y")
    N.check(N.lib().omni_vae_conv2d(C.byref(p), _stream()), "omni_vae_conv2d")
    return y if norm_gamma is None else (y, yn)


def vae_upsample2x_bordered(x):
    """Nearest-exact x2 upsample between zero-bordered rasters: [B,H+2,W+2,C] -> [B,2H+2,2W+2,C]."""
    B, Hp, Wp, Cc = x.shape
    y = torch.empty(B, 2 * (Hp - 2) + 2, 2 * (Wp - 2) + 2, Cc, dtype=BF16, device=x.device)
    N.check(N.lib().omni_vae_upsample2x_bordered(_p(x.contiguous(), name="x"), _p(y), B, Hp - 2, Wp - 2, Cc, _stream()),
            "omni_vae_upsample2x_bordered")
    return y


def vae_rmsnorm_silu(x, gamma, silu: bool = True):
    y = torch.empty_like(x)
    Cc = x.shape[-1]
    N.check(N.lib().omni_vae_rmsnorm_silu(_p(x.contiguous(), name="x"), _p(y), x.numel() // Cc, Cc, _p(gamma),
                                          int(silu), _stream()), "omni_vae_rmsnorm_silu")
    return y


VAE_ATTENTION_CHANNELS = 384          # the head dim omni_vae_attention is built for


def vae_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """Single-head attention per image: q, k, v [B, tokens, 384] (row-strided views of one fused projection are fine: the last
    dim must be contiguous, images must be tokens * row_stride apart) -> [B, tokens, 384] = softmax(scale q k^T) v."""
    B, tok, Cc = q.shape
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if t.shape != q.shape or t.stride(2) != 1 or (B > 1 and t.stride(0) != tok * t.stride(1)):
            raise N.OmniNativeError(f"{n}: expected [B, tokens, C] with contiguous channels and images tokens * row_stride apart")
    out = torch.empty(B, tok, Cc, dtype=BF16, device=q.device)
    N.check(N.lib().omni_vae_attention(_p(q, name="q"), _p(k, name="k"), _p(v, name="v"), _p(out), B, tok, Cc, q.stride(1),
                                       k.stride(1), v.stride(1), Cc, float(scale), _stream()), "omni_vae_attention")
    return out


def softmax_rows_(s: torch.Tensor, scale: float):
    rows, cols, ld = _rows2d(s, "scores")
    N.check(N.lib().omni_softmax_rows(_p(s, name="scores"), ld, rows, cols, scale, _stream()), "omni_softmax_rows")
    return s
