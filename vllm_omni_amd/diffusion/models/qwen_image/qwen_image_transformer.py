"""QwenImageTransformer2DModel on the CDNA4 kernels.

Mirror of the reference model class (vllm_omni/diffusion/models/qwen_image/qwen_image_transformer.py:609-839):
same constructor contract (`od_config`, widths as keyword defaults, only `num_layers` read from
`od_config.tf_model_config`), same parameter names (so diffusers / reference checkpoints load unchanged through
`load_weights`, including the q/k/v -> to_qkv and add_q/k/v -> add_kv_proj stacking of :805-815), same
`forward(hidden_states, encoder_hidden_states, encoder_hidden_states_mask, timestep, img_shapes, txt_seq_lens, ...)`
signature and `(sample,)` return.  The attributes TeaCache reaches for (`img_in, txt_norm, txt_in,
time_text_embed, transformer_blocks[i].img_mod, norm_out, proj_out, do_true_cfg`) exist under the same names.

What differs is HOW it computes: the module tree only owns parameters; `forward` hands raw device pointers to
`omni_dit_forward` (csrc/dit_forward.hip), which enqueues the fused kernel sequence on the current HIP stream.
There is no eager/PyTorch fallback: without libomni_cdna4.so or off-GPU, forward raises.
"""
from __future__ import annotations

import ctypes as C
from collections.abc import Iterable

import torch
import torch.nn as nn

import weakref

from ...batch import RaggedBatch, build_ragged_batch
from ...layers.adalayernorm import AdaLayerNorm
from .... import _native as N
from .... import ops
from .rope import grid_tokens, normalize_grids, rope_table

BF16 = torch.bfloat16


class _Param(nn.Module):
    """weight (+bias) holder; parameters are allocated uninitialised on `device` (no CPU staging of 20 B params).

    `forward(x)` = x @ W^T + b on the C-ABI kernels, so the modules the cache hooks call directly (`img_in`, `txt_in`,
    `proj_out`, the timestep-embedder linears: cache/teacache/extractors.py:189-245) behave like the reference's
    nn.Linear / ReplicatedLinear.  The eight big matrices of a block are `blockable`: the native runner may hold them in
    the K32-blocked layout, so they are only reachable through the block's own forward."""

    def __init__(self, out_features: int, in_features: int | None, bias: bool, device, dtype, blockable: bool = False):
        super().__init__()
        shape = (out_features,) if in_features is None else (out_features, in_features)
        self.weight = nn.Parameter(torch.empty(shape, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features, device=device, dtype=dtype), requires_grad=False) if bias else None
        self.blockable = blockable

    def forward(self, x: torch.Tensor, act_in: int = 0, act_out: int = 0) -> torch.Tensor:
        if self.weight.dim() != 2:
            raise TypeError("norm weight holder: not callable")
        if self.blockable:
            raise NotImplementedError("this projection lives inside the fused block kernels: call the block's forward")
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        if x2.shape[0] <= 8 and x2.shape[1] <= 4096:            # conditioning vectors: weight-streaming GEMV
            y = ops.linear_smallbatch(x2, self.weight, self.bias, act_in=act_in, act_out=act_out)
        else:
            if act_in or act_out:
                raise NotImplementedError("activations are fused only on the small-batch path")
            y = ops.linear(x2, self.weight, self.bias)
        return y.view(*x.shape[:-1], self.weight.shape[0])


class _RMSNormW(_Param):
    """vllm RMSNorm(hidden, eps) look-alike: y = x * rsqrt(mean(x^2) + eps) * weight."""

    def __init__(self, n, device, dtype, eps: float = 1e-6):
        super().__init__(n, None, False, device, dtype)
        self.variance_epsilon = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        return ops.rmsnorm(x2, self.weight, self.variance_epsilon).view(x.shape)


def _linear(i, o, device, dtype, blockable: bool = False):
    return _Param(o, i, True, device, dtype, blockable)


def _norm_w(n, device, dtype):
    return _RMSNormW(n, device, dtype)


class _TimestepEmbedder(nn.Module):
    def __init__(self, D, device, dtype):
        super().__init__()
        self.linear_1 = _linear(256, D, device, dtype)
        self.linear_2 = _linear(D, D, device, dtype)

    def forward(self, sample: torch.Tensor) -> torch.Tensor:
        """diffusers TimestepEmbedding: linear_1 -> SiLU -> linear_2."""
        out = []
        for i in range(0, sample.shape[0], 8):                  # the GEMV kernel takes <= 8 rows
            h = self.linear_1(sample[i:i + 8], act_out=1)
            out.append(self.linear_2(h))
        return torch.cat(out) if len(out) > 1 else out[0]


class QwenTimestepProjEmbeddings(nn.Module):
    """reference :40-62.  `use_additional_t_cond` (Layered variant): `addition_t_embedding = nn.Embedding(2, D)`, and the
    conditioning is timestep_emb + addition_t_embedding(additional_t_cond)."""

    def __init__(self, embedding_dim, device, dtype, use_additional_t_cond: bool = False):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedder(embedding_dim, device, dtype)
        self.use_additional_t_cond = use_additional_t_cond
        if use_additional_t_cond:
            self.addition_t_embedding = nn.Embedding(2, embedding_dim, device=device, dtype=dtype)

    def additional_rows(self, additional_t_cond, n: int):
        """[n, D] rows of addition_t_embedding for the native forward's `temb_add`, or None.  Mirrors the reference's checks."""
        if not self.use_additional_t_cond:
            if additional_t_cond is not None:
                raise ValueError("additional_t_cond was passed to a model built with use_additional_t_cond=False")
            return None
        if additional_t_cond is None:
            raise ValueError("When additional_t_cond is True, addition_t_cond must be provided.")     # reference :56-57
        idx = torch.as_tensor(additional_t_cond, dtype=torch.long, device=self.addition_t_embedding.weight.device).reshape(-1)
        if idx.numel() == 1 and n > 1:
            idx = idx.expand(n)
        if idx.numel() != n:
            raise ValueError(f"additional_t_cond carries {idx.numel()} entries for {n} conditioning rows")
        return self.addition_t_embedding.weight.detach()[idx].contiguous()

    def forward(self, timestep: torch.Tensor, hidden_states: torch.Tensor, additional_t_cond=None) -> torch.Tensor:
        proj = ops.timestep_sinusoid(timestep.to(torch.float32).contiguous(), 256, 1000.0)     # Timesteps(256, scale=1000)
        emb = self.timestep_embedder(proj)
        add = self.additional_rows(additional_t_cond, emb.shape[0])
        # module-level surface only (hooks that re-walk the model); inside omni_dit_forward this add is `temb_add`
        return emb if add is None else (emb + add)


class _GeluProj(nn.Module):
    def __init__(self, D, device, dtype):
        super().__init__()
        self.proj = _linear(D, 4 * D, device, dtype, blockable=True)


class _FeedForward(nn.Module):
    """diffusers FeedForward('gelu-approximate') parameter layout: net.0.proj, net.2."""

    def __init__(self, D, device, dtype):
        super().__init__()
        self.net = nn.ModuleList([_GeluProj(D, device, dtype), nn.Identity(), _linear(4 * D, D, device, dtype, blockable=True)])


class QwenImageCrossAttention(nn.Module):
    """Parameter holder for the joint attention of one block (reference :288-368); it runs inside the block kernels."""

    def __init__(self, D, head_dim, device, dtype):
        super().__init__()
        self.to_qkv = _linear(D, 3 * D, device, dtype, blockable=True)
        self.norm_q = _norm_w(head_dim, device, dtype)
        self.norm_k = _norm_w(head_dim, device, dtype)
        self.add_kv_proj = _linear(D, 3 * D, device, dtype, blockable=True)
        self.to_add_out = _linear(D, D, device, dtype, blockable=True)
        self.to_out = nn.ModuleList([_linear(D, D, device, dtype, blockable=True)])
        self.norm_added_q = _norm_w(head_dim, device, dtype)
        self.norm_added_k = _norm_w(head_dim, device, dtype)


class _Modulation(nn.Sequential):
    """`nn.Sequential(nn.SiLU(), Linear(D, 6D))` (reference :478-481) as ONE weight-streaming GEMV with the SiLU applied
    while the conditioning vector is staged (parameter names stay `img_mod.1.weight` / `.bias`)."""

    def __init__(self, D, device, dtype):
        super().__init__(nn.SiLU(), _linear(D, 6 * D, device, dtype))

    def forward(self, temb: torch.Tensor) -> torch.Tensor:
        out = [self[1](temb[i:i + 8], act_in=1) for i in range(0, temb.shape[0], 8)]
        return torch.cat(out) if len(out) > 1 else out[0]


class RotaryTables(tuple):
    """What `pos_embed(...)` returns: `(vid_freqs, txt_freqs)` complex64 like the reference (:222-285), plus the grid /
    text length they were built for so that a block called through the module surface can address the kernels' tables."""

    def __new__(cls, vid, txt, grid, txt_len):
        self = super().__new__(cls, (vid, txt))
        self.grid, self.txt_len = grid, txt_len
        return self


def _grid_of(img_shapes):
    """img_shapes as the pipelines pass it — [[(f, h, w)]] * B, or [[(f, h, w), (f2, h2, w2), ...]] * B for the Edit variants
    — to this build's grid: the reference reads img_shapes[0] only (:231-232), i.e. all items share item 0's shapes."""
    shp = img_shapes
    if isinstance(shp, (list, tuple)) and shp and isinstance(shp[0], (list, tuple)) and shp[0] and \
            isinstance(shp[0][0], (list, tuple)):
        shp = shp[0]                                       # [[entries...]] * B -> entries of item 0
    g = normalize_grids(tuple(tuple(int(v) for v in e) for e in shp) if isinstance(shp[0], (list, tuple)) else tuple(int(v) for v in shp))
    return g[0] if len(g) == 1 else g


class QwenEmbedRope(nn.Module):
    """`pos_embed` (reference :179-285): no parameters; tables are cached per (grid, text length)."""

    def forward(self, video_fhw, txt_seq_lens, device=None) -> RotaryTables:
        grid = _grid_of(video_fhw)
        T = int(max(txt_seq_lens)) if not isinstance(txt_seq_lens, int) else int(txt_seq_lens)
        cos, sin = rope_table(grid, T)
        cplx = torch.complex(cos, sin).to(device) if device is not None else torch.complex(cos, sin)
        return RotaryTables(cplx[T:], cplx[:T], grid, T)


class QwenEmbedLayer3DRope(QwenEmbedRope):
    """`pos_embed` of the Layered variant (reference :65-176): img_shapes = [(1, h, w)] * (layers + 1) + [(1, h_c, w_c)]; entry
    idx sits at frame idx, the LAST entry (the condition image) at frame -1, and the text positions start behind
    max(h/2, w/2, number of layers).  Same table layout and kernels: only the host-side index arithmetic differs
    (`rope.layered_grids`)."""

    def forward(self, video_fhw, txt_seq_lens, device=None) -> RotaryTables:
        from .rope import layered_grids

        grid = layered_grids(_grid_of(video_fhw))
        T = int(max(txt_seq_lens)) if not isinstance(txt_seq_lens, int) else int(txt_seq_lens)
        cos, sin = rope_table(grid, T)
        cplx = torch.complex(cos, sin).to(device) if device is not None else torch.complex(cos, sin)
        return RotaryTables(cplx[T:], cplx[:T], grid, T)


class QwenImageTransformerBlock(nn.Module):
    """One dual-stream MMDiT block (reference :461-605).  Inside `QwenImageTransformer2DModel.forward` the 60 blocks run in
    ONE native call; this module's own `forward` runs a single block through `omni_dit_block` with the reference's
    signature and `(encoder_hidden_states, hidden_states)` return, which is what cache hooks that re-walk the model need
    (cache/teacache/extractors.py:216-233), together with `img_mod`, `img_norm1` (:189-194)."""

    def __init__(self, D, head_dim, device, dtype):
        super().__init__()
        self.img_mod = _Modulation(D, device, dtype)
        self.img_norm1 = AdaLayerNorm(D, elementwise_affine=False, eps=1e-6)
        self.attn = QwenImageCrossAttention(D, head_dim, device, dtype)
        self.img_norm2 = AdaLayerNorm(D, elementwise_affine=False, eps=1e-6)
        self.img_mlp = _FeedForward(D, device, dtype)
        self.txt_mod = _Modulation(D, device, dtype)
        self.txt_norm1 = AdaLayerNorm(D, elementwise_affine=False, eps=1e-6)
        self.txt_norm2 = AdaLayerNorm(D, elementwise_affine=False, eps=1e-6)
        self.txt_mlp = _FeedForward(D, device, dtype)
        self.layer_idx = -1
        self._model_ref = None

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor,
                encoder_hidden_states_mask: torch.Tensor = None, temb: torch.Tensor = None,
                image_rotary_emb=None, joint_attention_kwargs=None, modulate_index=None):
        if modulate_index is not None:
            raise NotImplementedError(ZERO_COND_T_MESSAGE)
        model = self._model_ref() if self._model_ref is not None else None
        if model is None:
            raise RuntimeError("block is not attached to a QwenImageTransformer2DModel")
        return model._run_block(self.layer_idx, hidden_states, encoder_hidden_states, temb, image_rotary_emb)


class _NormOut(nn.Module):
    """diffusers AdaLayerNormContinuous(D, D, elementwise_affine=False, eps=1e-6): emb = linear(silu(c));
    scale, shift = emb.chunk(2); LN(x) * (1 + scale) + shift  (reference :686, :797)."""

    def __init__(self, D, device, dtype):
        super().__init__()
        self.linear = _linear(D, 2 * D, device, dtype)

    def forward(self, x: torch.Tensor, conditioning_embedding: torch.Tensor) -> torch.Tensor:
        B, S, D = x.shape
        emb = torch.cat([self.linear(conditioning_embedding[i:i + 8], act_in=1) for i in range(0, B, 8)]).contiguous()
        y = ops.adaln_modulate(x.reshape(B * S, D).contiguous(), emb, emb[:, D:], mod_item_stride=2 * D, rows_per_item=S)
        return y.view(B, S, D)


class Transformer2DModelOutput(tuple):
    """`(sample,)` with a `.sample` attribute (diffusers Transformer2DModelOutput look-alike)."""

    def __new__(cls, sample):
        return super().__new__(cls, (sample,))

    @property
    def sample(self):
        return self[0]


ZERO_COND_T_MESSAGE = (
    "zero_cond_t (the per-token `modulate_index` select between two modulation sets) is not built: in the reference snapshot "
    "the block accepts `modulate_index` but calls `self.img_norm1(hidden_states, img_mod1)` without it and multiplies a 2B-row "
    "`img_mod1` into B-row activations (qwen_image_transformer.py:552-564,590; its `_modulate` helper :505-539 is dead code), "
    "so there is no defined behaviour to match.  use_additional_t_cond and use_layer3d_rope (the other two Layered flags) ARE built.")


class QwenImageTransformer2DModel(nn.Module):
    def __init__(self, od_config=None, patch_size: int = 2, in_channels: int = 64, out_channels: int | None = 16,
                 num_layers: int = 60, attention_head_dim: int = 128, num_attention_heads: int = 24,
                 joint_attention_dim: int = 3584, guidance_embeds: bool = False,
                 axes_dims_rope: tuple[int, int, int] = (16, 56, 56), zero_cond_t: bool | None = None,
                 use_additional_t_cond: bool | None = None, use_layer3d_rope: bool | None = None, device=None, dtype=BF16):
        super().__init__()
        tfc = getattr(od_config, "tf_model_config", None) if od_config is not None else None
        if tfc is not None:
            nl = tfc.get("num_layers", None)
            num_layers = nl if nl is not None else num_layers
            dtype = getattr(od_config, "dtype", dtype)
        # the three Layered flags: explicit argument, else transformer/config.json (reference pipeline_qwen_image_layered.py:
        # 210-219 reads them from od_config.tf_model_config), else off
        flag = lambda v, k: bool(v if v is not None else (tfc.get(k, False) if tfc is not None else False))  # noqa: E731
        zero_cond_t, use_additional_t_cond = flag(zero_cond_t, "zero_cond_t"), flag(use_additional_t_cond, "use_additional_t_cond")
        use_layer3d_rope = flag(use_layer3d_rope, "use_layer3d_rope")
        if zero_cond_t:
            raise NotImplementedError(ZERO_COND_T_MESSAGE)
        self.zero_cond_t, self.use_additional_t_cond, self.use_layer3d_rope = False, use_additional_t_cond, use_layer3d_rope
        if attention_head_dim != 128 or tuple(axes_dims_rope) != (16, 56, 56):
            raise ValueError("the CDNA4 attention / RoPE kernels are built for head_dim 128, axes (16, 56, 56)")
        if dtype != BF16:
            raise ValueError("the CDNA4 DiT path computes in bf16 storage / fp32 accumulate only")
        device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.in_channels = in_channels
        self.out_channels = out_channels or in_channels
        self.patch_size = patch_size
        self.num_heads = num_attention_heads
        self.head_dim = attention_head_dim
        self.inner_dim = num_attention_heads * attention_head_dim
        self.joint_attention_dim = joint_attention_dim
        self.guidance_embeds = guidance_embeds
        self.do_true_cfg = False
        D = self.inner_dim
        self.time_text_embed = QwenTimestepProjEmbeddings(D, device, dtype, use_additional_t_cond)
        self.txt_norm = _norm_w(joint_attention_dim, device, dtype)
        self.img_in = _linear(in_channels, D, device, dtype)
        self.txt_in = _linear(joint_attention_dim, D, device, dtype)
        self.transformer_blocks = nn.ModuleList(
            [QwenImageTransformerBlock(D, attention_head_dim, device, dtype) for _ in range(num_layers)])
        self.norm_out = _NormOut(D, device, dtype)
        self.proj_out = _linear(D, patch_size * patch_size * self.out_channels, device, dtype)
        self.pos_embed = QwenEmbedLayer3DRope() if use_layer3d_rope else QwenEmbedRope()
        for i, blk in enumerate(self.transformer_blocks):
            blk.layer_idx, blk._model_ref = i, weakref.ref(self)
        self.teacache = None         # TeaCacheConfig when the native TeaCache path is enabled (cache/teacache/backend.py)
        self._native = None        # (DitWeights struct, keep-alive list)
        self._native_gen = 0       # bumped whenever the pointer table is invalidated (captured hipGraphs must be re-captured)
        self._w_blocked = False    # the 8 big matrices per layer currently hold the K32-blocked re-layout
        self.fp8 = False           # enable_fp8(): the block GEMMs run on e4m3 copies of their weights (BASELINE config 5)
        self._fp8_classes = frozenset()
        self._workspace = None
        self._batch_cache: dict = {}
        self._mod_tables: dict = {}   # modulation_table_for_schedule(): (weights generation, sigmas, t_cond) -> table, newest last

    # ------------------------------------------------------------------ weights
    @property
    def device(self):
        return self.proj_out.weight.device

    def _gemm_weight_params(self):
        """The eight [out, in] matrices per layer that the grouped GEMMs read (omni_dit_weights.gemm_w_k32_blocked)."""
        for blk in self.transformer_blocks:
            a = blk.attn
            yield from (a.to_qkv.weight, a.add_kv_proj.weight, a.to_out[0].weight, a.to_add_out.weight,
                        blk.img_mlp.net[0].proj.weight, blk.img_mlp.net[2].weight,
                        blk.txt_mlp.net[0].proj.weight, blk.txt_mlp.net[2].weight)

    def _invalidate_native(self) -> None:
        """Drop the cached raw-pointer table.  The generation counter moves NOW (not when the table is lazily rebuilt), so
        that anything holding device addresses of the old storages — a captured hipGraph in the pipeline — is seen as stale
        before it can be replayed."""
        self._native = None
        self._native_gen += 1

    FP8_CLASSES = ("qkv", "out", "mlp_up", "mlp_down")      # the four GEMM classes of a block (both streams each)

    def enable_fp8(self, on=True, classes=None) -> None:
        """Run block GEMMs in fp8 (OCP e4m3 operands on the scaled MFMA at twice the bf16 rate, omni_gemm_params.fp8): weights
        are quantised ONCE per output channel when the native pointer table is (re)built, activations per token in front of
        each GEMM; everything else stays bf16.  The bf16 parameters are kept (they remain the source of truth for state_dict /
        load_weights).  `classes` (ABI v9) picks which of the four GEMM classes run in fp8 — default: all; `on` may also be
        that tuple.  `FP8_RECIPE_ACCURATE` = the attention-side projections only: measured at 60 full-width layers
        (tools/fp8_error_budget.py) the MLP GEMMs carry 85 % of the fp8 error, and this recipe stays within
        1.6x of the bf16 path's own drift from fp32 where all-fp8 is at 3.9x.  BASELINE.json config 5; the reference has no fp8 path to match: accuracy is stated against the
        bf16 path / the fp32 oracle in tests/test_gpu_fp8.py and printed next to every fp8 throughput figure by bench.py."""
        if isinstance(on, (tuple, list, set, frozenset)):
            on, classes = True, on
        want = frozenset(self.FP8_CLASSES if classes is None else classes) if on else frozenset()
        if not want <= set(self.FP8_CLASSES):
            raise ValueError(f"fp8 classes {sorted(want)}: choose from {self.FP8_CLASSES}")
        if want != getattr(self, "_fp8_classes", frozenset()):
            self._fp8_classes = want
            self.fp8 = bool(want)
            self._workspace = None          # the workspace grows by the e4m3 activation buffer
            self._invalidate_native()

    FP8_RECIPE_ACCURATE = ("qkv", "out")

    def _set_weight_layout(self, blocked: bool) -> None:
        """In-place (one matrix of scratch) switch between the reference's row-major [out, in] and the K32-blocked order
        [in/32][out][32] the ring GEMM's LDS-DMA reads in whole cache lines (include/omni_cdna4.h).  Parameters are in
        row-major order whenever Python code can see them being written (init / load_weights); they are re-laid-out once,
        lazily, when the native pointer table is built."""
        if blocked == self._w_blocked:
            return
        for p in self._gemm_weight_params():
            n, k = p.shape
            if blocked:
                p.data = ops.w_to_k32_blocked(p.data)
            else:
                p.data = p.data.view(k // 32, n, 32).transpose(0, 1).contiguous().view(n, k)
        self._w_blocked = blocked
        self._invalidate_native()

    def unblock_weights(self) -> None:
        """Put every parameter back into the reference's row-major layout (e.g. before `state_dict()` / saving: after the
        first forward the eight GEMM matrices per layer hold the K32-blocked re-layout of the same values)."""
        self._set_weight_layout(False)

    # nn.Module plumbing that must never see (or strand) the K32-blocked re-layout / the cached raw-pointer table
    def state_dict(self, *args, **kwargs):
        """Always the reference's row-major values: the in-place blocked re-layout is undone first (it is re-applied
        lazily by the next forward)."""
        self.unblock_weights()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        self.unblock_weights()
        out = super().load_state_dict(state_dict, *args, **kwargs)
        self._invalidate_native()
        return out

    def _apply(self, fn, *args, **kwargs):
        """.to() / .cuda() / .half() move or replace parameter storage: drop every cached device pointer first."""
        self.unblock_weights()
        self._invalidate_native()
        self._workspace = None
        self._batch_cache.clear()
        return super()._apply(fn, *args, **kwargs)

    def init_random_(self, seed: int = 1234, std: float = 0.02) -> "QwenImageTransformer2DModel":
        """Synthetic weights ON DEVICE (bench only): >=2-D ~ N(0, std^2), biases 0, norm weights 1."""
        self._set_weight_layout(False)
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, p in self.named_parameters():
            if p.dim() >= 2:
                p.data.normal_(0.0, std, generator=g)
            elif "norm" in name:
                p.data.fill_(1.0)
            else:
                p.data.zero_()
        self._invalidate_native()
        return self

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        """Same contract as the reference loader (:804-839): HF names with split q/k/v are stacked into
        to_qkv / add_kv_proj (row order q|k|v); fused names are accepted as they are."""
        stacked = [(".to_qkv", ".to_q", 0), (".to_qkv", ".to_k", 1), (".to_qkv", ".to_v", 2),
                   (".add_kv_proj", ".add_q_proj", 0), (".add_kv_proj", ".add_k_proj", 1),
                   (".add_kv_proj", ".add_v_proj", 2)]
        self._set_weight_layout(False)
        params = dict(self.named_parameters())
        loaded: set[str] = set()
        D = self.inner_dim
        for name, w in weights:
            for fused, split, idx in stacked:
                # match ".to_q." exactly (".to_q" is a prefix of ".to_qkv": the reference's substring test would
                # mangle already-fused names)
                if (split + ".") in name:
                    name = name.replace(split + ".", fused + ".")
                    dst = params[name].data[idx * D:(idx + 1) * D]
                    if dst.shape != w.shape:
                        raise ValueError(f"{name}[{idx}]: expected {tuple(dst.shape)}, got {tuple(w.shape)}")
                    dst.copy_(w)
                    break
            else:
                if name not in params:
                    raise KeyError(f"unexpected weight {name}")
                if params[name].shape != w.shape:
                    raise ValueError(f"{name}: expected {tuple(params[name].shape)}, got {tuple(w.shape)}")
                params[name].data.copy_(w)
            loaded.add(name)
        self._invalidate_native()  # derived pointer tables must be rebuilt after loading (SURVEY.md §8b Ownership)
        return loaded

    def _native_weights(self) -> N.DitWeights:
        if self._native is not None:
            return self._native[0]
        self._set_weight_layout(True)
        for n, p in self.named_parameters():
            if not p.is_cuda or p.dtype != BF16 or not p.is_contiguous():
                raise N.OmniNativeError(f"parameter {n} must be a contiguous bf16 GPU tensor (got {p.device}, {p.dtype})")
        L = len(self.transformer_blocks)
        layers = (N.DitLayerWeights * L)()
        for i, blk in enumerate(self.transformer_blocks):
            a = blk.attn
            vals = dict(
                img_mod_w=blk.img_mod[1].weight, img_mod_b=blk.img_mod[1].bias,
                txt_mod_w=blk.txt_mod[1].weight, txt_mod_b=blk.txt_mod[1].bias,
                to_qkv_w=a.to_qkv.weight, to_qkv_b=a.to_qkv.bias, add_qkv_w=a.add_kv_proj.weight,
                add_qkv_b=a.add_kv_proj.bias, norm_q_w=a.norm_q.weight, norm_k_w=a.norm_k.weight,
                norm_added_q_w=a.norm_added_q.weight, norm_added_k_w=a.norm_added_k.weight,
                to_out_w=a.to_out[0].weight, to_out_b=a.to_out[0].bias, to_add_out_w=a.to_add_out.weight,
                to_add_out_b=a.to_add_out.bias,
                img_mlp_w1=blk.img_mlp.net[0].proj.weight, img_mlp_b1=blk.img_mlp.net[0].proj.bias,
                img_mlp_w2=blk.img_mlp.net[2].weight, img_mlp_b2=blk.img_mlp.net[2].bias,
                txt_mlp_w1=blk.txt_mlp.net[0].proj.weight, txt_mlp_b1=blk.txt_mlp.net[0].proj.bias,
                txt_mlp_w2=blk.txt_mlp.net[2].weight, txt_mlp_b2=blk.txt_mlp.net[2].bias)
            for k, v in vals.items():
                setattr(layers[i], k, v.data_ptr())
        w = N.DitWeights()
        w.num_layers, w.num_heads, w.head_dim = L, self.num_heads, self.head_dim
        w.joint_dim, w.in_channels = self.joint_attention_dim, self.in_channels
        w.out_channels_packed = self.proj_out.weight.shape[0]
        w.gemm_w_k32_blocked = 1 if self._w_blocked else 0
        te = self.time_text_embed.timestep_embedder
        w.t_lin1_w, w.t_lin1_b = te.linear_1.weight.data_ptr(), te.linear_1.bias.data_ptr()
        w.t_lin2_w, w.t_lin2_b = te.linear_2.weight.data_ptr(), te.linear_2.bias.data_ptr()
        w.txt_norm_w = self.txt_norm.weight.data_ptr()
        w.img_in_w, w.img_in_b = self.img_in.weight.data_ptr(), self.img_in.bias.data_ptr()
        w.txt_in_w, w.txt_in_b = self.txt_in.weight.data_ptr(), self.txt_in.bias.data_ptr()
        w.norm_out_w, w.norm_out_b = self.norm_out.linear.weight.data_ptr(), self.norm_out.linear.bias.data_ptr()
        w.proj_out_w, w.proj_out_b = self.proj_out.weight.data_ptr(), self.proj_out.bias.data_ptr()
        w.layers = C.cast(layers, C.POINTER(N.DitLayerWeights))
        keep = [layers]
        if self.fp8:
            f8 = (N.DitFp8Layer * L)()
            names = N._FP8_FIELDS
            for i, blk in enumerate(self.transformer_blocks):
                a = blk.attn
                mats = (a.to_qkv.weight, a.add_kv_proj.weight, a.to_out[0].weight, a.to_add_out.weight,
                        blk.img_mlp.net[0].proj.weight, blk.img_mlp.net[2].weight,
                        blk.txt_mlp.net[0].proj.weight, blk.txt_mlp.net[2].weight)
                cls_of = ("qkv", "qkv", "out", "out", "mlp_up", "mlp_down", "mlp_up", "mlp_down")
                for f, m, c in zip(names, mats, cls_of):
                    if c not in self._fp8_classes:
                        continue                                                              # NULL pointers: class stays bf16
                    w8, sc = ops.quantize_fp8_rows(m.data, x_k32_blocked=self._w_blocked)     # per output channel
                    keep += [w8, sc]
                    setattr(f8[i], f + ("_w8" if "mlp" not in f else "_8"), w8.data_ptr())
                    setattr(f8[i], f + "_s", sc.data_ptr())
            w.fp8_layers = C.cast(f8, C.POINTER(N.DitFp8Layer))
            keep.append(f8)
        self._native = (w, keep)
        return w

    # ------------------------------------------------------------------ batches
    def prepare_batch(self, batch: RaggedBatch) -> dict:
        """Upload the int32 maps and the bf16 RoPE table of a RaggedBatch (cache by content)."""
        key = (tuple(batch.txt_lens), tuple(batch.temb_rows), batch.grid, batch.txt_pos_end, batch.img_start, batch.s_img)
        hit = self._batch_cache.get(key)
        if hit is not None:
            return hit
        dev = self.device
        cos, sin = rope_table(batch.grid, batch.txt_pos_end)
        ent = dict(batch=batch, maps=batch.device_maps(dev), cos=cos.to(dev, BF16).contiguous(),
                   sin=sin.to(dev, BF16).contiguous())
        if len(self._batch_cache) > 32:
            self._batch_cache.clear()
        self._batch_cache[key] = ent
        return ent

    def _descriptor(self, prepared: dict, rows_img: int, rows_txt: int) -> tuple:
        """DitBatch with the batch maps, RoPE tables and workspace filled in (the caller adds its tensors)."""
        rb: RaggedBatch = prepared["batch"]
        lib = N.lib()
        w = self._native_weights()
        if rows_img != rb.n_img_rows or rows_txt != rb.n_txt_rows:
            raise ValueError(f"row counts do not match the batch descriptor: {rows_img} image / {rows_txt} text rows")
        need = lib.omni_dit_workspace_bytes(C.byref(w), rb.n_img_rows, rb.n_txt_rows, rb.n_temb)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        m = prepared["maps"]
        b = N.DitBatch()
        b.n_items, b.n_img_rows, b.n_txt_rows = rb.n_items, rb.n_img_rows, rb.n_txt_rows
        b.n_joint_rows, b.n_temb, b.max_seqlen = rb.n_joint_rows, rb.n_temb, rb.max_seqlen
        b.cu_seqlens, b.img_item, b.txt_item = m["cu_seqlens"].data_ptr(), m["img_item"].data_ptr(), m["txt_item"].data_ptr()
        b.img_joint_row, b.txt_joint_row = m["img_joint_row"].data_ptr(), m["txt_joint_row"].data_ptr()
        b.joint_pos, b.txt_pos_end = m["joint_pos"].data_ptr(), rb.txt_pos_end
        b.rope_cos, b.rope_sin = prepared["cos"].data_ptr(), prepared["sin"].data_ptr()
        b.workspace, b.workspace_bytes = self._workspace.data_ptr(), self._workspace.numel()
        return lib, w, b

    def forward_ragged(self, prepared: dict, latents: torch.Tensor, prompt_embeds: torch.Tensor,
                       timestep: torch.Tensor, out: torch.Tensor | None = None, teacache=None,
                       additional_t_cond=None, temb_add: torch.Tensor | None = None,
                       mod_table: torch.Tensor | None = None) -> torch.Tensor:
        """latents [n_img_rows, 64] bf16, prompt_embeds [n_txt_rows, joint_dim] bf16, timestep [n_temb] fp32
        (sigma = t/1000 exactly as the pipeline passes it) -> noise_pred [n_img_rows, 64] bf16.
        `teacache`: a cache.teacache.native.TeaCacheDeviceState for this batch (device-side decisions, no host sync)."""
        rb: RaggedBatch = prepared["batch"]
        if latents.dim() != 2 or prompt_embeds.dim() != 2 or latents.shape[1] != self.in_channels \
                or prompt_embeds.shape[1] != self.joint_attention_dim:
            raise ValueError(f"row counts do not match the batch descriptor: {tuple(latents.shape)}, {tuple(prompt_embeds.shape)}")
        lib, w, b = self._descriptor(prepared, latents.shape[0], prompt_embeds.shape[0])
        for t, dt, nm in ((latents, BF16, "latents"), (prompt_embeds, BF16, "prompt_embeds"), (timestep, torch.float32, "timestep")):
            if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
                raise N.OmniNativeError(f"{nm} must be a contiguous {dt} GPU tensor")
        if timestep.numel() != rb.n_temb:
            raise ValueError("timestep must have one entry per temb row")
        if out is None:
            out = torch.empty(rb.n_img_rows, w.out_channels_packed, dtype=BF16, device=self.device)
        b.latents, b.prompt_embeds, b.timestep = latents.data_ptr(), prompt_embeds.data_ptr(), timestep.data_ptr()
        b.noise_pred = out.data_ptr()
        if teacache is not None:
            b.teacache = C.pointer(teacache.struct_for(rb))
        # Layered variant: addition_t_embedding rows, one per conditioning row (`additional_t_cond`: ints, e.g. is_rgb = 0)
        # (`temb_add`: the same rows, gathered ONCE by a caller that replays the step as a hipGraph — the index tensor would be
        # a host-to-device copy inside the capture)
        if temb_add is None:
            temb_add = self.time_text_embed.additional_rows(additional_t_cond, rb.n_temb)
        elif temb_add.shape != (rb.n_temb, self.inner_dim) or temb_add.dtype != BF16 or not temb_add.is_contiguous():
            raise ValueError("temb_add must be a contiguous bf16 [n_temb, D] tensor")
        if temb_add is not None:
            b.temb_add = temb_add.data_ptr()
        if mod_table is not None:                            # this forward's rows of modulation_table(): the GEMVs are skipped
            L = len(self.transformer_blocks)
            if tuple(mod_table.shape) != (L, 2, rb.n_temb, 6 * self.inner_dim) or mod_table.dtype != BF16 \
                    or not mod_table.is_contiguous() or not mod_table.is_cuda:
                raise ValueError(f"mod_table must be a contiguous bf16 GPU tensor [{L}, 2, {rb.n_temb}, {6 * self.inner_dim}]")
            b.mod_table = mod_table.data_ptr()
        N.check(lib.omni_dit_forward(C.byref(w), C.byref(b), torch.cuda.current_stream().cuda_stream), "omni_dit_forward")
        return out

    @torch.no_grad()
    def modulation_table(self, sigma: torch.Tensor, additional_t_cond=None) -> torch.Tensor:
        """The blocks' modulation vectors for M conditioning rows in ONE pass over the 13.6 GB of modulation weights
        (omni_dit_modulation_table): sigma fp32 [M] (what forward_ragged takes as `timestep`, e.g. all the steps of a schedule)
        -> bf16 [num_layers, 2, M, 6 D].  A denoise loop hands `table[:, :, rows]` of its current step to forward_ragged instead
        of re-streaming those weights in every forward."""
        sig = sigma.to(self.device, torch.float32).reshape(-1).contiguous()
        M = int(sig.numel())
        temb = self.time_text_embed(sig, None, additional_t_cond).contiguous()                    # [M, D] bf16, the forward's own kernels
        lib, w = N.lib(), self._native_weights()
        L, D = len(self.transformer_blocks), self.inner_dim
        table = torch.empty(L, 2, M, 6 * D, dtype=BF16, device=self.device)
        need = lib.omni_dit_modulation_table_workspace_bytes(C.byref(w), M)
        ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        N.check(lib.omni_dit_modulation_table(C.byref(w), temb.data_ptr(), M, table.data_ptr(), ws.data_ptr(), need,
                                              torch.cuda.current_stream().cuda_stream), "omni_dit_modulation_table")
        return table

    MOD_TABLE_CACHE_BYTES = 1 << 30        # tables kept for later requests (a 50-step table is 221 MB; 288 GB of HBM)

    @torch.no_grad()
    def modulation_table_for_schedule(self, sigma: torch.Tensor, additional_t_cond=None, cache: bool = True,
                                      sigma_host: torch.Tensor | None = None) -> torch.Tensor:
        """`modulation_table`, remembered per SCHEDULE.  The table is a function of the weights and of the conditioning rows alone
        (the sigmas of the schedule, the Layered variant's additional_t_cond) — not of the prompt, the latents or the seed — and a
        server sees the same (resolution, step count) again and again: requests that share a schedule share the table, and the
        pass over the 13.6 GB of modulation weights (3.4 ms: 5 % of a 256x256 / 4-step image) is paid by the first of them
        only.  The reference streams these weights in every forward (qwen_image_transformer.py:552-561).  Keyed by the weights
        generation (a weight load / re-layout / fp8 switch drops every entry), the exact fp32 sigmas and t_cond; newest-first
        eviction beyond MOD_TABLE_CACHE_BYTES.  The returned table is shared: callers must not write to it.  `sigma_host`: the
        same values on the host (the key is formed from them: reading a device tensor back would wait for the stream)."""
        if not cache:
            return self.modulation_table(sigma, additional_t_cond)
        sig = (sigma if sigma_host is None else sigma_host).detach().to("cpu", torch.float32).reshape(-1)
        self._native_weights()                                   # (re)build the pointer table first: the generation is then current
        key = (self._native_gen, sig.numpy().tobytes(), None if additional_t_cond is None else tuple(int(t) for t in additional_t_cond))
        hit = self._mod_tables.pop(key, None)
        if hit is None:
            for k in [k for k in self._mod_tables if k[0] != self._native_gen]:
                del self._mod_tables[k]                          # tables of weights that no longer exist
            hit = self.modulation_table(sigma, additional_t_cond)
        self._mod_tables[key] = hit                              # newest last
        total = sum(t.numel() * 2 for t in self._mod_tables.values())
        for k in list(self._mod_tables):
            if total <= self.MOD_TABLE_CACHE_BYTES or k == key:
                break
            total -= self._mod_tables.pop(k).numel() * 2
        return hit

    def _run_block(self, layer: int, hidden_states, encoder_hidden_states, temb, image_rotary_emb):
        """Module-surface entry of ONE block (QwenImageTransformerBlock.forward): [B,S,D], [B,T,D], temb [B,D] -> (enc, hid)."""
        B, S, D = hidden_states.shape
        T = encoder_hidden_states.shape[1]
        grid = getattr(image_rotary_emb, "grid", None)
        if grid is None or grid_tokens(grid) != S:
            raise ValueError("image_rotary_emb must come from this model's pos_embed(img_shapes, txt_seq_lens)")
        prepared = self.prepare_batch(build_ragged_batch([T] * B, grid, txt_pos_end=max(T, getattr(image_rotary_emb, "txt_len", T))))
        hid = hidden_states.reshape(B * S, D).to(BF16).clone()
        enc = encoder_hidden_states.reshape(B * T, D).to(BF16).clone()
        tb = temb.to(BF16).contiguous()
        if tb.shape != (B, D):
            raise ValueError(f"temb must be [B, D], got {tuple(tb.shape)}")
        lib, w, b = self._descriptor(prepared, B * S, B * T)
        N.check(lib.omni_dit_block(C.byref(w), layer, C.byref(b), hid.data_ptr(), enc.data_ptr(), tb.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), "omni_dit_block")
        return enc.view(B, T, D), hid.view(B, S, D)

    # ------------------------------------------------------------------ Ulysses sequence parallelism
    def _sp_forward_gen(self, rank: int, P: int, latents: torch.Tensor, prompt_embeds: torch.Tensor, sigma: torch.Tensor,
                        grid, t_cond: int | None = None, teacache=None):
        """One DiT forward of ONE item with its image tokens split over P ranks (reference qwen_image_transformer.py:
        735-742,776-781,800-801 + attention/parallel/ulysses.py:59-135), written as a generator that YIELDS its collectives
        (`("all_to_all", send[P, ...])` -> recv, `("all_gather", x)` -> [P, ...], `("all_reduce", x)` -> sum) so that the same
        code is driven by torch.distributed (forward_sp) or, in tests, by an in-process exchange between P generators on one GPU.

        latents [S, 64] (full; this rank takes rows [rank*S/P, ...)), prompt_embeds [T, joint] (replicated), sigma fp32 [1].
        `grid`: one (f, h, w) triple, or SEVERAL (the Edit pipelines: target + condition images on one sequence axis, reference
        pipeline_qwen_image_edit.py:600-632; the Layered variant's explicit frame indices) — S is then the total over all of
        them and the rows are sharded as ONE sequence, exactly as the reference chunks `hidden_states` after the concatenation.
        `t_cond`: the Layered variant's additional_t_cond.  Per block: omni_dit_block_qkv on the local rows -> ONE fused
        all-to-all of the image q/k/v (sequence <-> heads; the replicated text q/k/v are only head-sliced, no communication —
        unlike the reference, which sends P copies of the text queries through the all-to-all) -> flash attention over the whole
        sequence for H/P heads -> ONE all-to-all back (image rows to their owners + the text rows' head slice to everyone) ->
        omni_dit_block_post.

        `teacache`: a cache.teacache.sp_state.TeaCacheSPState of this (rank, item) — TeaCache under sequence parallelism.  The
        decision input (relative L1 distance of consecutive modulated inputs of block 0, reference hook.py:195-206) is a mean over
        ALL image rows: every rank sums its slice, ONE all-reduce of two floats makes the sums global, and every rank takes the
        same decision — the single-device decision.  The cached residual is this rank's row slice."""
        H, d, D = self.num_heads, self.head_dim, self.inner_dim
        S = grid_tokens(grid)
        if S % P or H % P:
            raise ValueError(f"Ulysses needs the image-token count ({S}) and the heads ({H}) divisible by the degree ({P})")
        if latents.shape[0] != S:
            raise ValueError(f"{latents.shape[0]} latent rows for a token grid of {S}")
        S_loc, Hh, T = S // P, H // P, prompt_embeds.shape[0]
        dev = self.device
        rb = build_ragged_batch([T], grid, img_rows=(rank * S_loc, S_loc))
        prepared = self.prepare_batch(rb)
        lib = N.lib()
        stream = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
        hid = self.img_in(latents[rank * S_loc:(rank + 1) * S_loc].to(dev, BF16)).contiguous()          # [S_loc, D]
        enc = self.txt_in(self.txt_norm(prompt_embeds.to(dev, BF16))).contiguous()                      # [T, D]
        temb = self.time_text_embed(sigma.to(dev), hid, None if t_cond is None else [t_cond]).contiguous()   # [1, D]
        compute, hid_in = True, None
        if teacache is not None:
            first = self.transformer_blocks[0]
            img_mod1, _ = first.img_mod(temb).chunk(2, dim=-1)
            mod, _ = first.img_norm1(hid.view(1, S_loc, D), img_mod1)
            mod = mod.reshape(S_loc, D).contiguous()
            if teacache.cnt > 0 and teacache.prev_mod is not None:
                # the reference subtracts two bf16 tensors (rounded to bf16) before abs().mean() (hook.py:197-201)
                part = torch.stack([(mod - teacache.prev_mod).abs().float().sum(), teacache.prev_mod.abs().float().sum()])
                total = yield ("all_reduce", part)
                compute = teacache.decide(total, S * D)
            else:
                teacache.first()
            teacache.prev_mod = mod
            if compute or teacache.prev_res is None:
                compute, hid_in = True, hid.clone()
            else:
                hid = (hid + teacache.prev_res).contiguous()
        if compute:
            cu = torch.tensor([0, T + S], dtype=torch.int32, device=dev)
            scale = 1.0 / (d ** 0.5)
            for l in range(len(self.transformer_blocks)):
                _, w, b = self._descriptor(prepared, S_loc, T)
                ws_base, ws_t = self._workspace.data_ptr(), self._workspace
                qp, kp, vp = C.c_void_p(), C.c_void_p(), C.c_void_p()
                N.check(lib.omni_dit_block_qkv(C.byref(w), l, C.byref(b), hid.data_ptr(), enc.data_ptr(), temb.data_ptr(),
                                               C.byref(qp), C.byref(kp), C.byref(vp), stream()), "omni_dit_block_qkv")
                view = lambda p: ws_t[p.value - ws_base: p.value - ws_base + (T + S_loc) * D * 2].view(BF16).view(T + S_loc, H, d)  # noqa: E731
                q, k, v = view(qp), view(kp), view(vp)                       # joint order: [text ; image chunk]
                # my head slice of the replicated text rows (copied: the workspace is reused by the next native call)
                hs = slice(rank * Hh, (rank + 1) * Hh)
                txt_qkv = torch.stack([q[:T, hs], k[:T, hs], v[:T, hs]]).reshape(3, T, Hh * d).clone()
                send = torch.stack([q[T:], k[T:], v[T:]]).view(3, S_loc, P, Hh * d).permute(2, 0, 1, 3).contiguous()
                recv = yield ("all_to_all", send)                            # [P(source = sequence chunk), 3, S_loc, Hh*d]
                img_qkv = recv.permute(1, 0, 2, 3).reshape(3, S, Hh * d)
                full = torch.cat([txt_qkv, img_qkv], dim=1).contiguous()     # [3, T + S, Hh*d]
                o = ops.flash_attn_varlen(full[0], full[1], full[2], cu, Hh, T + S, scale)           # [T + S, Hh*d]
                send2 = torch.cat([o[T:].view(P, S_loc, Hh * d), o[:T].unsqueeze(0).expand(P, T, Hh * d)], dim=1).contiguous()
                recv2 = yield ("all_to_all", send2)                          # [P(source = head slice), S_loc + T, Hh*d]
                loc = recv2.permute(1, 0, 2).reshape(S_loc + T, D)           # heads in source-rank order = original order
                attn = torch.cat([loc[S_loc:], loc[:S_loc]]).contiguous()    # back to the joint order [text ; image chunk]
                _, w, b = self._descriptor(prepared, S_loc, T)
                N.check(lib.omni_dit_block_post(C.byref(w), l, C.byref(b), hid.data_ptr(), enc.data_ptr(), temb.data_ptr(),
                                                attn.data_ptr(), stream()), "omni_dit_block_post")
            if teacache is not None:
                teacache.prev_res = hid - hid_in                             # this rank's rows of the residual (hook.py:151)
        out_loc = self.proj_out(self.norm_out(hid.view(1, S_loc, D), temb)).view(S_loc, -1)
        gathered = yield ("all_gather", out_loc.contiguous())            # [P, S_loc, 64]
        return gathered.reshape(S, -1)

    @torch.no_grad()
    def forward_sp(self, latents: torch.Tensor, prompt_embeds: torch.Tensor, sigma: torch.Tensor,
                   grid, group=None) -> torch.Tensor:
        """Ulysses sequence-parallel forward over `group` (RCCL): every rank passes the same full `latents` / prompt and
        receives the full noise prediction [S_img, 64].  240 all-to-alls + 1 all-gather per 60-layer forward."""
        return self.forward_sp_multi([(latents, prompt_embeds, sigma)], grid, group)[0]

    @torch.no_grad()
    def forward_sp_multi(self, items: list[tuple[torch.Tensor, torch.Tensor, torch.Tensor]], grid, group=None, *,
                         t_cond: int | None = None, teacache_states=None, emulate_ranks: int = 0) -> list[torch.Tensor]:
        """Several sequence-parallel forwards over the same grid — the two true-CFG branches of a request, or several
        requests — software-pipelined: while one forward's all-to-all is in flight the next forward's GEMMs run
        (distributed/sp_driver.py).  items[i] = (latents [S, 64], prompt_embeds [T_i, joint], sigma fp32 [1]).
        `teacache_states[i]`: the TeaCacheSPState of item i on this rank (None: no TeaCache).
        `emulate_ranks = P` (tests, one device): P virtual ranks run the same forwards with an in-process exchange
        (`teacache_states[r][i]` then belongs to virtual rank r); every rank's result is checked to be identical."""
        import torch.distributed as dist

        from ...distributed.sp_driver import drive, drive_in_process

        if emulate_ranks:
            P = int(emulate_ranks)
            gens = [[self._sp_forward_gen(r, P, lat, pe, sg, grid, t_cond,
                                          None if teacache_states is None else teacache_states[r][i])
                     for i, (lat, pe, sg) in enumerate(items)] for r in range(P)]
            return drive_in_process(gens)
        P = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        gens = [self._sp_forward_gen(rank, P, lat, pe, sg, grid, t_cond, None if teacache_states is None else teacache_states[i])
                for i, (lat, pe, sg) in enumerate(items)]
        return drive(gens, group)

    # ------------------------------------------------------------------ reference-shaped forward
    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                encoder_hidden_states_mask: torch.Tensor = None, timestep: torch.Tensor = None,
                img_shapes=None, txt_seq_lens=None, guidance=None, attention_kwargs=None,
                additional_t_cond=None, return_dict: bool = True):
        """Reference signature (:692-802).  B items of equal text length T (as the reference batches them)."""
        if guidance is not None:
            raise NotImplementedError("guidance-distilled variants are outside the Qwen-Image path")
        B, S_img, _ = hidden_states.shape
        T = encoder_hidden_states.shape[1]
        grid = _grid_of(img_shapes)
        if self.use_layer3d_rope:
            from .rope import layered_grids

            grid = layered_grids(grid)
        if grid_tokens(grid) != S_img:
            raise ValueError(f"img_shapes {grid} does not match {S_img} image tokens")
        prepared = self.prepare_batch(build_ragged_batch([T] * B, grid))
        # the reference casts timestep to the activation dtype before the sinusoid (:746)
        ts = timestep.to(device=self.device, dtype=hidden_states.dtype).to(torch.float32).contiguous()
        out = self.forward_ragged(prepared, hidden_states.reshape(B * S_img, -1).contiguous(),
                                  encoder_hidden_states.reshape(B * T, -1).contiguous(), ts,
                                  additional_t_cond=additional_t_cond)
        out = out.view(B, S_img, -1)
        return Transformer2DModelOutput(out) if return_dict else (out,)
