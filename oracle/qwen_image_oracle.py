"""TEST INFRASTRUCTURE — the CPU oracle for the Qwen-Image DiT denoising path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module, and only as the checker.  The product path (vllm_omni_amd/) never does.

This is a from-scratch *restatement* (functional torch, fp32 by default, CPU) of the
algorithm the reference runs on the hot path.  Every function cites the reference
file:line it follows (paths relative to /root/reference/).  Parameters are passed as a
flat `dict[str, Tensor]` keyed by the reference's own parameter names
(`transformer_blocks.{i}.attn.to_qkv.weight`, ...), so the same seeded weights drive
the oracle, the shim-imported reference (oracle/gen_golden.py) and the HIP path.

Pinning: the reference holds NO golden vectors for this path, so the
oracle is pinned against *outputs of the reference itself run in the authoring
container* — oracle/gen_golden.py shim-imports the unmodified reference DiT
(oracle/ref_shims.py), runs it on seeded inputs and commits the results under
tests/golden/.  tests/test_oracle_golden.py checks this oracle against those fixtures.
The third-party leaf ops (vllm RMSNorm/linears, diffusers FeedForward/Timesteps/
AdaLayerNormContinuous/FlowMatchEulerDiscreteScheduler/VAE) are absent from
/root/reference and from this image; their arithmetic is restated from their published
definitions and is **parity-unpinned** against the real packages (vllm v0.12.0,
diffusers >= 0.36.0).  The VAE decode and the scheduler have no reference-run fixture
either (diffusers is not importable here): for them this file is the spec.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

Params = dict  # dict[str, torch.Tensor]


# =============================================================================== DiT leaf ops
def timestep_sinusoid(t: torch.Tensor, dim: int = 256, scale: float = 1000.0) -> torch.Tensor:
    """Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0, scale=1000).

    qwen_image_transformer.py:44; body duplicated in-tree at pipeline_qwen_image.py:135-184.
    Output order after the flip is [cos | sin].
    """
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = scale * emb
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def timestep_embedding(P: Params, t: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """QwenTimestepProjEmbeddings.forward — qwen_image_transformer.py:50-62 (no additional_t_cond)."""
    proj = timestep_sinusoid(t).to(dtype)
    h = F.linear(proj, P["time_text_embed.timestep_embedder.linear_1.weight"],
                 P["time_text_embed.timestep_embedder.linear_1.bias"])
    h = F.silu(h)
    return F.linear(h, P["time_text_embed.timestep_embedder.linear_2.weight"],
                    P["time_text_embed.timestep_embedder.linear_2.bias"])


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """vllm RMSNorm native path (call sites qwen_image_transformer.py:397-400,758)."""
    dt = x.dtype
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return xf.to(dt) * weight


def layer_norm_noaffine(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def ada_layer_norm(x: torch.Tensor, mod: torch.Tensor, eps: float = 1e-6):
    """AdaLayerNorm.forward_native — diffusion/layers/adalayernorm.py:94-102 (index=None).

    mod [B, 3D] chunks as (shift, scale, gate); returns (LN(x)*(1+scale)+shift, gate[B,1,D]).
    """
    shift, scale, gate = mod.chunk(3, dim=-1)
    return layer_norm_noaffine(x, eps) * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1), gate.unsqueeze(1)


def rope_interleaved(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """RotaryEmbedding(is_neox_style=False).forward_native — diffusion/layers/rope.py:12-36,140-151.

    x [B,S,H,dh]; cos/sin [S, dh/2]:  out[2i] = x[2i]c - x[2i+1]s ; out[2i+1] = x[2i+1]c + x[2i]s
    """
    c = cos.repeat_interleave(2, dim=-1)[None, :, None, :]
    s = sin.repeat_interleave(2, dim=-1)[None, :, None, :]
    x1, x2 = x[..., 0::2], x[..., 1::2]
    rot = torch.stack((-x2, x1), dim=-1).flatten(-2)
    return x * c + rot * s


FUSED_SDPA = False   # True: call F.scaled_dot_product_attention — the very op the reference's SDPAImpl calls (sdpa.py:55-63).  The
                     # CHECKER keeps the spelled-out softmax(QK^T)V below (no dependence on a fused kernel's internals); bench.py's
                     # cpu_baseline sets this so that the timed port does the reference's work the reference's way
                     # (tools/time_reference_cpu.py: port / reference seconds per block on the authoring host).


def sdpa_nhd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """SDPAImpl.forward — diffusion/attention/backends/sdpa.py:46-66: non-causal, no mask. [B,S,H,dh]."""
    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    if FUSED_SDPA:
        return F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=scale).permute(0, 2, 1, 3)
    p = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
    return (p @ v).permute(0, 2, 1, 3)


def sdpa_nhd_general(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, attn_mask: torch.Tensor | None = None,
                     is_causal: bool = False) -> torch.Tensor:
    """SDPAImpl.forward with everything it forwards to F.scaled_dot_product_attention — diffusion/attention/backends/sdpa.py:46-66:
    `attn_mask=attn_metadata.attn_mask`, `is_causal=self.causal`, `scale=self.softmax_scale`, q [B,Sq,H,dh], k / v [B,Sk,Hkv,dh]
    (grouped K / V heads are repeated, as torch's enable_gqa does).  Spelled out (the documented semantics of the torch op, no
    fused kernel): bool mask True = attend, float mask added to the scaled scores, is_causal = lower triangle aligned top-left.
    A row without any attendable key yields zeros here (torch's math path: NaN; its fused paths: zeros)."""
    B, Sq, H, dh = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    q, k, v = (t.permute(0, 2, 1, 3).float() for t in (q, k, v))
    if Hkv != H:
        k, v = k.repeat_interleave(H // Hkv, dim=1), v.repeat_interleave(H // Hkv, dim=1)
    s = (q @ k.transpose(-1, -2)) * scale
    if is_causal:
        s = s.masked_fill(~torch.ones(Sq, Sk, dtype=torch.bool, device=s.device).tril(), float("-inf"))
    if attn_mask is not None:
        s = s.masked_fill(~attn_mask, float("-inf")) if attn_mask.dtype == torch.bool else s + attn_mask.float()
    dead = torch.isinf(s).all(dim=-1, keepdim=True)
    p = torch.softmax(s.masked_fill(dead, 0.0), dim=-1).masked_fill(dead, 0.0)
    return (p @ v).permute(0, 2, 1, 3)


def feed_forward(P: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """diffusers FeedForward('gelu-approximate') = Linear -> GELU(tanh) -> Linear (call :491,501,591,596)."""
    h = F.gelu(F.linear(x, P[pre + ".net.0.proj.weight"], P[pre + ".net.0.proj.bias"]), approximate="tanh")
    return F.linear(h, P[pre + ".net.2.weight"], P[pre + ".net.2.bias"])


# =============================================================================== RoPE tables
def rope_axis_freqs(index: torch.Tensor, dim: int, theta: float = 10000.0):
    """QwenEmbedRope.rope_params — qwen_image_transformer.py:207-220; returns (cos, sin) [len, dim/2]."""
    inv = 1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float32).div(dim))
    ang = torch.outer(index.to(torch.float32), inv)
    return torch.cos(ang), torch.sin(ang)


def rope_tables_multi(grids, txt_len: int, axes_dim=(16, 56, 56), theta: float = 10000.0):
    """QwenEmbedRope.forward for a LIST of (frame, height, width) entries (the Edit pipelines: target + condition images on
    one sequence axis) — qwen_image_transformer.py:236-262: entry idx starts at frame position idx; the text positions start
    behind the largest half-extent of any entry."""
    cos, sin, start = [], [], 0
    for idx, (f, h, w) in enumerate(grids):
        (vc, vs), _ = rope_tables(f, h, w, 1, axes_dim, theta, frame_offset=idx)
        cos.append(vc)
        sin.append(vs)
        start = max(start, h // 2, w // 2)
    t_idx = torch.arange(start, start + txt_len)
    tc = torch.cat([rope_axis_freqs(t_idx, d, theta)[0] for d in axes_dim], dim=1)
    ts = torch.cat([rope_axis_freqs(t_idx, d, theta)[1] for d in axes_dim], dim=1)
    return (torch.cat(cos), torch.cat(sin)), (tc, ts)


def rope_tables_layered(grids, txt_len: int, axes_dim=(16, 56, 56), theta: float = 10000.0):
    """QwenEmbedLayer3DRope.forward (Layered variant) — qwen_image_transformer.py:101-142: entries idx < last sit at frame
    position idx (`_compute_video_freqs(.., idx)`, :144-161), the LAST entry is the condition image at frame position -1
    (`_compute_condition_freqs`, :163-176: `freqs_neg[0][-1:]`); text positions start at max(h//2, w//2 over all entries,
    layer_num = len(grids) - 1) (:131-138)."""
    cos, sin, start = [], [], 0
    layer_num = len(grids) - 1
    for idx, (f, h, w) in enumerate(grids):
        (vc, vs), _ = rope_tables(f, h, w, 1, axes_dim, theta, frame_offset=(idx if idx != layer_num else -1))
        cos.append(vc)
        sin.append(vs)
        start = max(start, h // 2, w // 2)
    start = max(start, layer_num)
    t_idx = torch.arange(start, start + txt_len)
    tc = torch.cat([rope_axis_freqs(t_idx, d, theta)[0] for d in axes_dim], dim=1)
    ts = torch.cat([rope_axis_freqs(t_idx, d, theta)[1] for d in axes_dim], dim=1)
    return (torch.cat(cos), torch.cat(sin)), (tc, ts)


def rope_tables(frame: int, height: int, width: int, txt_len: int, axes_dim=(16, 56, 56), theta: float = 10000.0,
                frame_offset: int = 0):
    """QwenEmbedRope.forward + _compute_video_freqs, scale_rope=True — qwen_image_transformer.py:222-285.

    Returns (vid_cos, vid_sin) [f*h*w, 64] and (txt_cos, txt_sin) [txt_len, 64] in fp32.
    h/w indices are centred: [-(h - h//2) .. -1, 0 .. h//2 - 1]; text positions start at
    max(h//2, w//2) and use the same index on all three axes (:251-257).
    """
    def axis(idx, d):
        return rope_axis_freqs(idx, d, theta)

    f_idx = torch.arange(frame_offset, frame_offset + frame)
    h_idx = torch.cat([torch.arange(-(height - height // 2), 0), torch.arange(0, height // 2)])
    w_idx = torch.cat([torch.arange(-(width - width // 2), 0), torch.arange(0, width // 2)])
    fc, fs = axis(f_idx, axes_dim[0])
    hc, hs = axis(h_idx, axes_dim[1])
    wc, ws = axis(w_idx, axes_dim[2])

    def grid(a, b, c):
        a = a.view(frame, 1, 1, -1).expand(frame, height, width, -1)
        b = b.view(1, height, 1, -1).expand(frame, height, width, -1)
        c = c.view(1, 1, width, -1).expand(frame, height, width, -1)
        return torch.cat([a, b, c], dim=-1).reshape(frame * height * width, -1).contiguous()

    vid_cos, vid_sin = grid(fc, hc, wc), grid(fs, hs, ws)
    start = max(height // 2, width // 2)
    t_idx = torch.arange(start, start + txt_len)
    tc = torch.cat([axis(t_idx, d)[0] for d in axes_dim], dim=1)
    ts = torch.cat([axis(t_idx, d)[1] for d in axes_dim], dim=1)
    return (vid_cos, vid_sin), (tc, ts)


# =============================================================================== DiT block / forward
def joint_attention(P: Params, pre: str, img: torch.Tensor, txt: torch.Tensor, vid_cs, txt_cs,
                    num_heads: int, taps: dict | None = None):
    """QwenImageCrossAttention.forward — qwen_image_transformer.py:370-458 (non-SP branch)."""
    B, S_img, D = img.shape
    T = txt.shape[1]
    dh = D // num_heads
    qkv_i = F.linear(img, P[pre + ".to_qkv.weight"], P[pre + ".to_qkv.bias"])
    qkv_t = F.linear(txt, P[pre + ".add_kv_proj.weight"], P[pre + ".add_kv_proj.bias"])
    iq, ik, iv = (t.unflatten(-1, (num_heads, dh)) for t in qkv_i.chunk(3, dim=-1))
    tq, tk, tv = (t.unflatten(-1, (num_heads, dh)) for t in qkv_t.chunk(3, dim=-1))
    iq, ik = rms_norm(iq, P[pre + ".norm_q.weight"]), rms_norm(ik, P[pre + ".norm_k.weight"])
    tq, tk = rms_norm(tq, P[pre + ".norm_added_q.weight"]), rms_norm(tk, P[pre + ".norm_added_k.weight"])
    # cos/sin are cast to the activation dtype BEFORE rotating (:403-406)
    ic, isn = vid_cs[0].to(iq.device, iq.dtype), vid_cs[1].to(iq.device, iq.dtype)   # (device: the GPU tests run this oracle in fp32 ON the GPU as the checker)
    tc, tsn = txt_cs[0].to(tq.device, tq.dtype), txt_cs[1].to(tq.device, tq.dtype)
    iq, ik = rope_interleaved(iq, ic, isn), rope_interleaved(ik, ic, isn)
    tq, tk = rope_interleaved(tq, tc, tsn), rope_interleaved(tk, tc, tsn)
    q = torch.cat([tq, iq], dim=1)  # joint order [text ; image] (:412-416)
    k = torch.cat([tk, ik], dim=1)
    v = torch.cat([tv, iv], dim=1)
    o = sdpa_nhd(q, k, v, 1.0 / math.sqrt(dh)).flatten(2, 3)
    if taps is not None:
        taps.update(q=q, k=k, v=v, attn=o)
    txt_o, img_o = o[:, :T], o[:, T:]
    img_o = F.linear(img_o, P[pre + ".to_out.0.weight"], P[pre + ".to_out.0.bias"])
    txt_o = F.linear(txt_o, P[pre + ".to_add_out.weight"], P[pre + ".to_add_out.bias"])
    return img_o, txt_o


def dit_block(P: Params, i: int, hidden: torch.Tensor, enc: torch.Tensor, temb: torch.Tensor, vid_cs, txt_cs,
              num_heads: int, taps: dict | None = None):
    """QwenImageTransformerBlock.forward — qwen_image_transformer.py:541-605 (zero_cond_t=False)."""
    pre = f"transformer_blocks.{i}"
    st = F.silu(temb)
    img_mod = F.linear(st, P[pre + ".img_mod.1.weight"], P[pre + ".img_mod.1.bias"])
    txt_mod = F.linear(st, P[pre + ".txt_mod.1.weight"], P[pre + ".txt_mod.1.bias"])
    img_mod1, img_mod2 = img_mod.chunk(2, dim=-1)
    txt_mod1, txt_mod2 = txt_mod.chunk(2, dim=-1)
    img_n, img_g1 = ada_layer_norm(hidden, img_mod1)
    txt_n, txt_g1 = ada_layer_norm(enc, txt_mod1)
    if taps is not None:
        taps.update(img_mod=img_mod, txt_mod=txt_mod, img_n1=img_n, txt_n1=txt_n)
    img_a, txt_a = joint_attention(P, pre + ".attn", img_n, txt_n, vid_cs, txt_cs, num_heads, taps)
    hidden = hidden + img_g1 * img_a
    enc = enc + txt_g1 * txt_a
    img_n2, img_g2 = ada_layer_norm(hidden, img_mod2)
    hidden = hidden + img_g2 * feed_forward(P, pre + ".img_mlp", img_n2)
    txt_n2, txt_g2 = ada_layer_norm(enc, txt_mod2)
    enc = enc + txt_g2 * feed_forward(P, pre + ".txt_mlp", txt_n2)
    if enc.dtype == torch.float16:  # :600-603
        enc, hidden = enc.clip(-65504, 65504), hidden.clip(-65504, 65504)
    return enc, hidden


def num_layers_of(P: Params) -> int:
    n = 0
    while f"transformer_blocks.{n}.img_mod.1.weight" in P:
        n += 1
    return n


def dit_forward(P: Params, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor,
                timestep: torch.Tensor, img_shape: tuple[int, int, int], num_heads: int = 24,
                taps: dict | None = None, layer3d_rope: bool = False, additional_t_cond: torch.Tensor | None = None) -> torch.Tensor:
    """QwenImageTransformer2DModel.forward — qwen_image_transformer.py:692-802 (no SP, no guidance).

    hidden_states [B,S_img,64] packed latents, encoder_hidden_states [B,T,joint_dim],
    timestep [B] (= sigma; the pipeline passes t/1000, Timesteps(scale=1000) multiplies back),
    img_shape = (1, H/16, W/16).  Returns noise_pred [B,S_img,64].
    """
    T = encoder_hidden_states.shape[1]
    hidden = F.linear(hidden_states, P["img_in.weight"], P["img_in.bias"])
    timestep = timestep.to(hidden.dtype)
    enc = rms_norm(encoder_hidden_states, P["txt_norm.weight"])
    enc = F.linear(enc, P["txt_in.weight"], P["txt_in.bias"])
    temb = timestep_embedding(P, timestep, hidden.dtype)
    if additional_t_cond is not None:                    # Layered: conditioning = timesteps_emb + addition_t_embedding(c)  (:55-60)
        add = P["time_text_embed.addition_t_embedding.weight"][additional_t_cond.long().to(temb.device)]
        temb = temb + add.to(temb.dtype)
    if layer3d_rope:                                     # Layered: QwenEmbedLayer3DRope (:65-176)
        shp = [tuple(g) for g in img_shape] if isinstance(img_shape[0], (tuple, list)) else [tuple(img_shape)]
        vid_cs, txt_cs = rope_tables_layered(shp, T)
    elif isinstance(img_shape[0], (tuple, list)):        # several images on the sequence axis (Edit pipelines)
        vid_cs, txt_cs = rope_tables_multi([tuple(g) for g in img_shape], T)
    else:
        vid_cs, txt_cs = rope_tables(*img_shape, T)
    if taps is not None:
        taps.update(temb=temb, hidden_in=hidden, enc_in=enc)
    for i in range(num_layers_of(P)):
        bt = {} if (taps is not None) else None
        enc, hidden = dit_block(P, i, hidden, enc, temb, vid_cs, txt_cs, num_heads, bt)
        if taps is not None:
            bt.update(hidden=hidden, enc=enc)
            taps[f"block{i}"] = bt
    # AdaLayerNormContinuous (diffusers): emb = linear(silu(temb)); scale, shift = chunk(2)  (:797)
    emb = F.linear(F.silu(temb).to(hidden.dtype), P["norm_out.linear.weight"], P["norm_out.linear.bias"])
    scale, shift = emb.chunk(2, dim=1)
    hidden = layer_norm_noaffine(hidden) * (1 + scale)[:, None, :] + shift[:, None, :]
    return F.linear(hidden, P["proj_out.weight"], P["proj_out.bias"])  # :798


# =============================================================================== pipeline helpers
def pack_latents(latents: torch.Tensor) -> torch.Tensor:
    """QwenImagePipeline._pack_latents — pipeline_qwen_image.py:436-441. [B,C,H,W] -> [B,(H/2)(W/2),4C]."""
    B, C, H, W = latents.shape
    x = latents.view(B, C, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, (H // 2) * (W // 2), C * 4)


def unpack_latents(latents: torch.Tensor, height: int, width: int, vae_scale_factor: int = 8) -> torch.Tensor:
    """QwenImagePipeline._unpack_latents — pipeline_qwen_image.py:444-457. -> [B,C,1,H/8,W/8]."""
    B, _, ch = latents.shape
    h = 2 * (int(height) // (vae_scale_factor * 2))
    w = 2 * (int(width) // (vae_scale_factor * 2))
    x = latents.view(B, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(B, ch // 4, 1, h, w)


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    """pipeline_qwen_image.py:63-73."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


@dataclass
class SchedulerConfig:
    """Qwen-Image scheduler_config.json values (recalled, verify against a checkpoint)."""
    num_train_timesteps: int = 1000
    base_image_seq_len: int = 256
    max_image_seq_len: int = 8192
    base_shift: float = 0.5
    max_shift: float = 0.9
    shift_terminal: float | None = 0.02
    shift: float = 1.0


def flow_match_sigmas_mu(sigmas_in, mu: float, cfg: SchedulerConfig = SchedulerConfig()):
    """FlowMatchEulerDiscreteScheduler.set_timesteps(sigmas=, mu=) as the Layered pipeline drives it
    (pipeline_qwen_image_layered.py:808-816: sigmas = linspace(1, 0, N + 1)[:-1], mu = sqrt(S_cond / 256)): exponential shift
    with the GIVEN mu, terminal stretch, trailing 0.  Same float32 numpy arithmetic as `flow_match_sigmas`."""
    import numpy as np

    sigmas = np.array(sigmas_in).astype(np.float32)
    sigmas = math.exp(mu) / (math.exp(mu) + (1 / sigmas - 1) ** 1.0)
    if cfg.shift_terminal:
        one_minus = 1 - sigmas
        scale = one_minus[-1] / (1 - cfg.shift_terminal)
        sigmas = 1 - (one_minus / scale)
    sig = torch.from_numpy(np.asarray(sigmas)).to(torch.float32)
    return sig * cfg.num_train_timesteps, torch.cat([sig, torch.zeros(1)])


def layered_sigmas(num_inference_steps: int, cond_seq_len: int, cfg: SchedulerConfig = SchedulerConfig()):
    """pipeline_qwen_image_layered.py:808-816."""
    import numpy as np

    base_seqlen = 256 * 256 / 16 / 16
    return flow_match_sigmas_mu(np.linspace(1.0, 0, num_inference_steps + 1)[:-1], (cond_seq_len / base_seqlen) ** 0.5, cfg)


def layered_calculate_dimensions(target_area: float, ratio: float) -> tuple[int, int]:
    """pipeline_qwen_image_layered.py:108-116 -> (width, height), multiples of 32."""
    width = math.sqrt(target_area * ratio)
    height = width / ratio
    return round(width / 32) * 32, round(height / 32) * 32


def layered_pack_latents(latents: torch.Tensor) -> torch.Tensor:
    """QwenImageLayeredPipeline._pack_latents — pipeline_qwen_image_layered.py:518-524.  [B, L, C, H, W] -> [B, L*(H/2)(W/2), 4C]."""
    B, L, C, H, W = latents.shape
    x = latents.view(B, L, C, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4, 6)
    return x.reshape(B, L * (H // 2) * (W // 2), C * 4)


def layered_unpack_latents(latents: torch.Tensor, height: int, width: int, layers: int, vae_scale_factor: int = 8) -> torch.Tensor:
    """QwenImageLayeredPipeline._unpack_latents — :526-541.  -> [B, C, layers + 1, H/8, W/8]."""
    B, _, ch = latents.shape
    h = 2 * (int(height) // (vae_scale_factor * 2))
    w = 2 * (int(width) // (vae_scale_factor * 2))
    x = latents.view(B, layers + 1, h // 2, w // 2, ch // 4, 2, 2).permute(0, 1, 4, 2, 5, 3, 6)
    return x.reshape(B, layers + 1, ch // 4, h, w).permute(0, 2, 1, 3, 4)


def layered_diffuse(P: Params, latents: torch.Tensor, image_latents: torch.Tensor, prompt_embeds: torch.Tensor,
                    negative_prompt_embeds: torch.Tensor | None, img_shapes, num_inference_steps: int,
                    true_cfg_scale: float = 4.0, cfg_normalize: bool = False, num_heads: int = 24,
                    sched: SchedulerConfig = SchedulerConfig(), trajectory: list | None = None, is_rgb: int = 0) -> torch.Tensor:
    """QwenImageLayeredPipeline.diffuse — pipeline_qwen_image_layered.py:543-615: the generated frames and the condition image
    on one sequence axis, prediction sliced back to the generated rows, additional_t_cond = is_rgb, true-CFG combination with
    the norm rescale only when `cfg_normalize` (default off, :662)."""
    timesteps, sigmas = layered_sigmas(num_inference_steps, image_latents.shape[1], sched)
    do_cfg = negative_prompt_embeds is not None and true_cfg_scale > 1
    cond = torch.full((latents.shape[0],), int(is_rgb), dtype=torch.long)
    S = latents.shape[1]
    for i, t in enumerate(timesteps):
        ts = t.expand(latents.shape[0]).to(latents.dtype)
        x = torch.cat([latents, image_latents], dim=1)
        pred = dit_forward(P, x, prompt_embeds, ts / 1000, img_shapes, num_heads, layer3d_rope=True, additional_t_cond=cond)[:, :S]
        if do_cfg:
            neg = dit_forward(P, x, negative_prompt_embeds, ts / 1000, img_shapes, num_heads, layer3d_rope=True,
                              additional_t_cond=cond)[:, :S]
            pred = cfg_combine(pred, neg, true_cfg_scale) if cfg_normalize else neg + true_cfg_scale * (pred - neg)
        latents = euler_step(latents, pred, float(sigmas[i]), float(sigmas[i + 1]))
        if trajectory is not None:
            trajectory.append(latents.clone())
    return latents


def flow_match_sigmas(num_inference_steps: int, image_seq_len: int, cfg: SchedulerConfig = SchedulerConfig()):
    """prepare_timesteps (pipeline_qwen_image.py:492-508) + FlowMatchEulerDiscreteScheduler.set_timesteps
    (diffusers; use_dynamic_shifting=True, time_shift_type='exponential', shift_terminal).

    Returns (timesteps[N] fp32, sigmas[N+1] fp32 with trailing 0).  float32 numpy arithmetic, as in diffusers
    (pinned bit-exactly to tests/golden/pipe_helpers.npz = the reference's prepare_timesteps over the restated
    scheduler stub of oracle/ref_shims_pipeline.py; the diffusers class itself is absent: parity unpinned).
    """
    import numpy as np

    sigmas = np.linspace(1.0, 1.0 / num_inference_steps, num_inference_steps)
    mu = calculate_shift(image_seq_len, cfg.base_image_seq_len, cfg.max_image_seq_len, cfg.base_shift, cfg.max_shift)
    # diffusers: `sigmas = np.array(sigmas).astype(np.float32)` first, then float32 numpy arithmetic throughout
    sigmas = np.array(sigmas).astype(np.float32)
    sigmas = math.exp(mu) / (math.exp(mu) + (1 / sigmas - 1) ** 1.0)  # exponential time shift
    if cfg.shift_terminal:
        one_minus = 1 - sigmas
        scale = one_minus[-1] / (1 - cfg.shift_terminal)
        sigmas = 1 - (one_minus / scale)
    sig = torch.from_numpy(np.asarray(sigmas)).to(torch.float32)
    timesteps = sig * cfg.num_train_timesteps
    return timesteps, torch.cat([sig, torch.zeros(1)])


def euler_step(sample: torch.Tensor, model_output: torch.Tensor, sigma: float, sigma_next: float) -> torch.Tensor:
    """FlowMatchEulerDiscreteScheduler.step: fp32 update, cast back (call pipeline_qwen_image.py:585)."""
    prev = sample.to(torch.float32) + (sigma_next - sigma) * model_output.to(torch.float32)
    return prev.to(model_output.dtype)


def cfg_combine(pos: torch.Tensor, neg: torch.Tensor, scale: float) -> torch.Tensor:
    """True-CFG combine with norm rescale — pipeline_qwen_image.py:580-583."""
    comb = neg + scale * (pos - neg)
    cond_norm = torch.norm(pos, dim=-1, keepdim=True)
    noise_norm = torch.norm(comb, dim=-1, keepdim=True)
    return comb * (cond_norm / noise_norm)


def diffuse(P: Params, latents: torch.Tensor, prompt_embeds: torch.Tensor, negative_prompt_embeds: torch.Tensor | None,
            img_shape, num_inference_steps: int, true_cfg_scale: float = 4.0, num_heads: int = 24,
            sched: SchedulerConfig = SchedulerConfig(), trajectory: list | None = None) -> torch.Tensor:
    """QwenImagePipeline.diffuse — pipeline_qwen_image.py:530-586.  latents [B,S_img,64]."""
    timesteps, sigmas = flow_match_sigmas(num_inference_steps, latents.shape[1], sched)
    do_cfg = negative_prompt_embeds is not None and true_cfg_scale > 1
    for i, t in enumerate(timesteps):
        ts = t.expand(latents.shape[0]).to(latents.dtype)
        pred = dit_forward(P, latents, prompt_embeds, ts / 1000, img_shape, num_heads)
        if do_cfg:
            neg = dit_forward(P, latents, negative_prompt_embeds, ts / 1000, img_shape, num_heads)
            pred = cfg_combine(pred, neg, true_cfg_scale)
        latents = euler_step(latents, pred, float(sigmas[i]), float(sigmas[i + 1]))
        if trajectory is not None:
            trajectory.append(latents.clone())
    return latents


# =============================================================================== TeaCache (optional cache accelerator)
TEACACHE_QWEN_COEFFICIENTS = [-4.5e02, 2.8e02, -4.5e01, 3.2e00, -2.0e-02]   # cache/teacache/config.py:18-27


class TeaCacheBranchState:
    """cache/teacache/state.py:13-37."""

    def __init__(self):
        self.cnt, self.acc = 0, 0.0
        self.prev_mod = self.prev_res = None


def teacache_forward(P: Params, st: TeaCacheBranchState, hidden_states, encoder_hidden_states, timestep, img_shape,
                     num_heads: int, rel_l1_thresh: float, coefficients=TEACACHE_QWEN_COEFFICIENTS, log: list | None = None):
    """ONE hooked transformer forward: TeaCacheHook.new_forward (cache/teacache/hook.py:82-157) over extract_qwen_context
    (extractors.py:145-261) with the decision rule of _should_compute_full_transformer (hook.py:170-217).  `st` is the
    state of the CFG branch this forward belongs to (hook.py:113-121)."""
    import numpy as np

    T = encoder_hidden_states.shape[1]
    hidden = F.linear(hidden_states, P["img_in.weight"], P["img_in.bias"])                        # extractors.py:189-193
    timestep = timestep.to(hidden.dtype)
    enc = F.linear(rms_norm(encoder_hidden_states, P["txt_norm.weight"]), P["txt_in.weight"], P["txt_in.bias"])
    temb = timestep_embedding(P, timestep, hidden.dtype)
    vid_cs, txt_cs = rope_tables(*img_shape, T)
    img_mod = F.linear(F.silu(temb), P["transformer_blocks.0.img_mod.1.weight"], P["transformer_blocks.0.img_mod.1.bias"])
    modulated, _ = ada_layer_norm(hidden, img_mod.chunk(2, dim=-1)[0])                            # extractors.py:208-211
    # --- decision (hook.py:188-217)
    if st.cnt == 0:
        st.acc, compute = 0.0, True
    elif st.prev_mod is None:
        compute = True
    else:
        rel = float(((modulated - st.prev_mod).abs().mean() / (st.prev_mod.abs().mean() + 1e-8)).item())
        st.acc += abs(float(np.poly1d(coefficients)(rel)))
        if st.acc < rel_l1_thresh:
            compute = False
        else:
            st.acc, compute = 0.0, True
    if log is not None:
        log.append(compute)
    if not compute and st.prev_res is not None:                                                   # hook.py:127-134
        out = hidden + st.prev_res
    else:                                                                                         # hook.py:135-157
        h, e = hidden, enc
        for i in range(num_layers_of(P)):
            e, h = dit_block(P, i, h, e, temb, vid_cs, txt_cs, num_heads)
        st.prev_res = h - hidden
        out = h
    st.prev_mod = modulated
    st.cnt += 1
    emb = F.linear(F.silu(temb).to(out.dtype), P["norm_out.linear.weight"], P["norm_out.linear.bias"])   # postprocess
    scale, shift = emb.chunk(2, dim=1)
    out = layer_norm_noaffine(out) * (1 + scale)[:, None, :] + shift[:, None, :]
    return F.linear(out, P["proj_out.weight"], P["proj_out.bias"])


def teacache_diffuse(P: Params, latents, prompt_embeds, negative_prompt_embeds, img_shape, num_inference_steps: int,
                     rel_l1_thresh: float, true_cfg_scale: float = 4.0, num_heads: int = 24,
                     sched: SchedulerConfig = SchedulerConfig(), trajectory: list | None = None):
    """The reference `diffuse` loop (pipeline_qwen_image.py:530-586) over the TeaCache-hooked transformer: forwards
    alternate positive / negative, each branch with its own state.  Returns (final latents, compute log per branch)."""
    timesteps, sigmas = flow_match_sigmas(num_inference_steps, latents.shape[1], sched)
    pos_st, neg_st = TeaCacheBranchState(), TeaCacheBranchState()
    log_p, log_n = [], []
    for i, t in enumerate(timesteps):
        ts = t.expand(latents.shape[0]).to(latents.dtype)
        p = teacache_forward(P, pos_st, latents, prompt_embeds, ts / 1000, img_shape, num_heads, rel_l1_thresh, log=log_p)
        n = teacache_forward(P, neg_st, latents, negative_prompt_embeds, ts / 1000, img_shape, num_heads, rel_l1_thresh, log=log_n)
        latents = euler_step(latents, cfg_combine(p, n, true_cfg_scale), float(sigmas[i]), float(sigmas[i + 1]))
        if trajectory is not None:
            trajectory.append(latents.clone())
    return latents, (log_p, log_n)


# =============================================================================== prompt encoding (request-side boundary)
def qwen_prompt_embeds(hidden: torch.Tensor, attention_mask: torch.Tensor, drop_idx: int = 34):
    """`_extract_masked_hidden` + the rest of `_get_qwen_prompt_embeds` after the text-encoder call
    (pipeline_qwen_image.py:351-357, 378-392): per-sample valid rows, minus the `drop_idx` template tokens, zero-padded to
    the longest, with the matching mask.  hidden [B, L, D] = text_encoder(...).hidden_states[-1]."""
    mask = attention_mask.bool()
    lens = mask.sum(dim=1).tolist()
    split = [e[drop_idx:] for e in torch.split(hidden[mask], lens, dim=0)]
    T = max(e.size(0) for e in split)
    emb = torch.stack([torch.cat([u, u.new_zeros(T - u.size(0), u.size(1))]) for u in split])
    msk = torch.stack([torch.cat([torch.ones(u.size(0), dtype=torch.long), torch.zeros(T - u.size(0), dtype=torch.long)])
                       for u in split])
    return emb, msk


def encode_prompt_postprocess(emb: torch.Tensor, msk: torch.Tensor, num_images_per_prompt: int, max_sequence_length: int):
    """encode_prompt after `_get_qwen_prompt_embeds` (pipeline_qwen_image.py:423-433): truncate, repeat per image."""
    emb, msk = emb[:, :max_sequence_length], msk[:, :max_sequence_length]
    B, T, _ = emb.shape
    emb = emb.repeat(1, num_images_per_prompt, 1).view(B * num_images_per_prompt, T, -1)
    msk = msk.repeat(1, num_images_per_prompt, 1).view(B * num_images_per_prompt, T)
    return emb, msk


# =============================================================================== VAE decode (T = 1)
LATENTS_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
                0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
LATENTS_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
               3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]
# autoencoder_kl_qwenimage.py:679-697


@dataclass
class VaeConfig:
    base_dim: int = 96
    z_dim: int = 16
    dim_mult: tuple = (1, 2, 4, 4)
    num_res_blocks: int = 2
    temporal_upsample: tuple = (True, True, False)  # reversed temperal_downsample [F,T,T] (:700)
    latents_mean: list = field(default_factory=lambda: list(LATENTS_MEAN))
    latents_std: list = field(default_factory=lambda: list(LATENTS_STD))

    def decoder_dims(self):
        return [self.base_dim * u for u in [self.dim_mult[-1]] + list(self.dim_mult[::-1])]  # :590


def _conv3d_as_2d(P: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    """QwenImageCausalConv3d on a single frame with an empty cache — autoencoder_kl_qwenimage.py:69-84.

    Causal padding puts 2 zero frames in FRONT, so for T=1 only temporal kernel slice [-1] touches
    data; the 3-D conv equals a 2-D conv with weight[:, :, -1].  x is [B,C,H,W].
    """
    w = P[name + ".weight"]
    k = w.shape[-1]
    return F.conv2d(x, w[:, :, -1], P[name + ".bias"], padding=k // 2)


def vae_rms_norm(x: torch.Tensor, gamma: torch.Tensor) -> torch.Tensor:
    """QwenImageRMS_norm (channel_first) — :108-109: F.normalize(x, dim=1) * sqrt(C) * gamma."""
    return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) * gamma.reshape(1, -1, 1, 1)


def vae_res_block(P: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """QwenImageResidualBlock.forward — :252-285."""
    h = _conv3d_as_2d(P, pre + ".conv_shortcut", x) if (pre + ".conv_shortcut.weight") in P else x
    x = F.silu(vae_rms_norm(x, P[pre + ".norm1.gamma"]))
    x = _conv3d_as_2d(P, pre + ".conv1", x)
    x = F.silu(vae_rms_norm(x, P[pre + ".norm2.gamma"]))
    x = _conv3d_as_2d(P, pre + ".conv2", x)
    return x + h


def vae_attn_block(P: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """QwenImageAttentionBlock.forward — :305-330: single-head SDPA over H*W tokens, C channels."""
    B, C, H, W = x.shape
    idn = x
    x = vae_rms_norm(x, P[pre + ".norm.gamma"])
    qkv = F.conv2d(x, P[pre + ".to_qkv.weight"], P[pre + ".to_qkv.bias"])
    qkv = qkv.reshape(B, 1, C * 3, H * W).permute(0, 1, 3, 2)
    q, k, v = qkv.chunk(3, dim=-1)
    p = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(C), dim=-1)
    o = (p @ v).squeeze(1).permute(0, 2, 1).reshape(B, C, H, W)
    o = F.conv2d(o, P[pre + ".proj.weight"], P[pre + ".proj.bias"])
    return o + idn


def vae_upsample(P: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """QwenImageResample(upsample2d/3d) for a first chunk of T=1 — :170-176 ('Rep' skips time_conv),
    then nearest-exact x2 in fp32 (:123-124) + Conv2d(dim, dim//2, 3, padding=1) (:148-157)."""
    up = F.interpolate(x.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").type_as(x)
    return F.conv2d(up, P[pre + ".resample.1.weight"], P[pre + ".resample.1.bias"], padding=1)


def vae_decode(P: Params, z: torch.Tensor, cfg: VaeConfig = VaeConfig(), clamp: bool = True) -> torch.Tensor:
    """AutoencoderKLQwenImage._decode for one frame — :839-863 + QwenImageDecoder3d.forward :618-664.

    z [B, z_dim, 1, h, w] (already de-normalised) -> image [B, 3, 1, 8h, 8w] clamped to [-1, 1].
    Parameter names are the vendored module's: post_quant_conv.*, decoder.conv_in.*, decoder.mid_block.*,
    decoder.up_blocks.{i}.resnets.{j}.*, decoder.up_blocks.{i}.upsamplers.0.resample.1.*, decoder.norm_out.gamma,
    decoder.conv_out.*
    """
    x = z[:, :, 0]
    x = _conv3d_as_2d(P, "post_quant_conv", x)
    x = _conv3d_as_2d(P, "decoder.conv_in", x)
    x = vae_res_block(P, "decoder.mid_block.resnets.0", x)
    x = vae_attn_block(P, "decoder.mid_block.attentions.0", x)
    x = vae_res_block(P, "decoder.mid_block.resnets.1", x)
    n_up = len(cfg.dim_mult)
    for i in range(n_up):
        for j in range(cfg.num_res_blocks + 1):
            x = vae_res_block(P, f"decoder.up_blocks.{i}.resnets.{j}", x)
        if i != n_up - 1:
            x = vae_upsample(P, f"decoder.up_blocks.{i}.upsamplers.0", x)
    x = F.silu(vae_rms_norm(x, P["decoder.norm_out.gamma"]))
    x = _conv3d_as_2d(P, "decoder.conv_out", x)
    return (torch.clamp(x, -1.0, 1.0) if clamp else x).unsqueeze(2)


# ---- spatial tiling (od_config.vae_use_tiling -> vae.use_tiling, registry.py:88-92) --------------------------------------------
def vae_blend_v(a: torch.Tensor, b: torch.Tensor, blend_extent: int) -> torch.Tensor:
    """AutoencoderKLQwenImage.blend_v — :889-895: the top `blend_extent` rows of b become a linear cross-fade from a's bottom
    rows, IN PLACE on b (the reference's tiles are modified in place, so a tile that serves as `a` later is already blended)."""
    blend_extent = min(a.shape[-2], b.shape[-2], blend_extent)
    for y in range(blend_extent):
        b[:, :, :, y, :] = a[:, :, :, -blend_extent + y, :] * (1 - y / blend_extent) + b[:, :, :, y, :] * (y / blend_extent)
    return b


def vae_blend_h(a: torch.Tensor, b: torch.Tensor, blend_extent: int) -> torch.Tensor:
    """AutoencoderKLQwenImage.blend_h — :897-903."""
    blend_extent = min(a.shape[-1], b.shape[-1], blend_extent)
    for x in range(blend_extent):
        b[:, :, :, :, x] = a[:, :, :, :, -blend_extent + x] * (1 - x / blend_extent) + b[:, :, :, :, x] * (x / blend_extent)
    return b


def _vae_blend_rows(rows, blend_h: int, blend_w: int, stride_h: int, stride_w: int) -> torch.Tensor:
    """The stitching loop shared by tiled_encode / tiled_decode — :955-968, :1014-1028: every tile is blended with the
    (already blended) tile above and the one to its left, cropped to the stride, rows concatenated."""
    result_rows = []
    for i, row in enumerate(rows):
        result_row = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = vae_blend_v(rows[i - 1][j], tile, blend_h)
            if j > 0:
                tile = vae_blend_h(row[j - 1], tile, blend_w)
            result_row.append(tile[:, :, :, :stride_h, :stride_w])
        result_rows.append(torch.cat(result_row, dim=-1))
    return torch.cat(result_rows, dim=3)


def vae_tiled_decode(P: Params, z: torch.Tensor, cfg: VaeConfig = VaeConfig(), tile_sample_min: int = 256,
                     tile_sample_stride: int = 192) -> torch.Tensor:
    """AutoencoderKLQwenImage.tiled_decode for one frame — :971-1031.  Latent tiles of (256/8)^2 every 192/8 positions are decoded
    independently (post_quant_conv + decoder), overlaps of 64 pixels are cross-faded, every tile contributes its first 192 x 192
    pixels.  NOTE the reference returns the stitched image UN-CLAMPED: `_decode` leaves for `tiled_decode` before its
    torch.clamp (:844-845 vs :857), so values outside [-1, 1] survive until the image processor."""
    _, _, _, height, width = z.shape
    sr = 8
    tmin, tstr = tile_sample_min // sr, tile_sample_stride // sr
    blend = tile_sample_min - tile_sample_stride
    rows = []
    for i in range(0, height, tstr):
        row = []
        for j in range(0, width, tstr):
            row.append(vae_decode(P, z[:, :, :, i:i + tmin, j:j + tmin], cfg, clamp=False))
        rows.append(row)
    dec = _vae_blend_rows(rows, blend, blend, tile_sample_stride, tile_sample_stride)
    return dec[:, :, :, : height * sr, : width * sr]


def vae_encode_moments(P: Params, image: torch.Tensor, cfg: VaeConfig = VaeConfig()) -> torch.Tensor:
    """encoder + quant_conv of one frame: all 2 * z_dim channels (mean | logvar) — the tensor tiled_encode blends (:948-949)."""
    x = _conv3d_as_2d(P, "encoder.conv_in", image[:, :, 0])
    dims = [cfg.base_dim * u for u in [1] + list(cfg.dim_mult)]
    k = 0
    for i in range(len(dims) - 1):
        for _ in range(cfg.num_res_blocks):
            x = vae_res_block(P, f"encoder.down_blocks.{k}", x)
            k += 1
        if i != len(cfg.dim_mult) - 1:
            x = vae_downsample(P, f"encoder.down_blocks.{k}", x)
            k += 1
    x = vae_res_block(P, "encoder.mid_block.resnets.0", x)
    x = vae_attn_block(P, "encoder.mid_block.attentions.0", x)
    x = vae_res_block(P, "encoder.mid_block.resnets.1", x)
    x = F.silu(vae_rms_norm(x, P["encoder.norm_out.gamma"]))
    x = _conv3d_as_2d(P, "encoder.conv_out", x)
    return _conv3d_as_2d(P, "quant_conv", x).unsqueeze(2)


def vae_tiled_encode(P: Params, image: torch.Tensor, cfg: VaeConfig = VaeConfig(), tile_sample_min: int = 256,
                     tile_sample_stride: int = 192) -> torch.Tensor:
    """AutoencoderKLQwenImage.tiled_encode for one frame + DiagonalGaussianDistribution.mode() — :905-969: image tiles of 256^2
    every 192 pixels through encoder + quant_conv, 8-latent-row overlaps cross-faded, posterior mean [B, z_dim, 1, H/8, W/8]."""
    _, _, _, height, width = image.shape
    sr = 8
    tmin, tstr = tile_sample_min // sr, tile_sample_stride // sr
    blend = tmin - tstr
    rows = []
    for i in range(0, height, tile_sample_stride):
        row = []
        for j in range(0, width, tile_sample_stride):
            row.append(vae_encode_moments(P, image[:, :, :, i:i + tile_sample_min, j:j + tile_sample_min], cfg))
        rows.append(row)
    enc = _vae_blend_rows(rows, blend, blend, tstr, tstr)[:, :, :, : height // sr, : width // sr]
    return enc[:, : cfg.z_dim]


def vae_downsample(P: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """QwenImageResample(downsample2d/3d) for a first chunk of T=1 — :162-166 (ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2));
    the 3-D variant's time_conv only touches LATER chunks (:201-211: the first chunk just fills the cache)."""
    return F.conv2d(F.pad(x, (0, 1, 0, 1)), P[pre + ".resample.1.weight"], P[pre + ".resample.1.bias"], stride=2)


def vae_encode(P: Params, image: torch.Tensor, cfg: VaeConfig = VaeConfig()) -> torch.Tensor:
    """AutoencoderKLQwenImage._encode for one frame + DiagonalGaussianDistribution.mode() — :788-835,
    QwenImageEncoder3d.forward :430-477.  image [B, 3, 1, H, W] in [-1, 1] -> posterior mean [B, z_dim, 1, H/8, W/8]
    (what the Edit pipelines take with sample_mode="argmax", pipeline_qwen_image_edit.py:459-467)."""
    x = _conv3d_as_2d(P, "encoder.conv_in", image[:, :, 0])
    dims = [cfg.base_dim * u for u in [1] + list(cfg.dim_mult)]
    k = 0
    for i, (i_dim, o_dim) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(cfg.num_res_blocks):
            x = vae_res_block(P, f"encoder.down_blocks.{k}", x)
            k += 1
        if i != len(cfg.dim_mult) - 1:
            x = vae_downsample(P, f"encoder.down_blocks.{k}", x)
            k += 1
    x = vae_res_block(P, "encoder.mid_block.resnets.0", x)
    x = vae_attn_block(P, "encoder.mid_block.attentions.0", x)
    x = vae_res_block(P, "encoder.mid_block.resnets.1", x)
    x = F.silu(vae_rms_norm(x, P["encoder.norm_out.gamma"]))
    x = _conv3d_as_2d(P, "encoder.conv_out", x)
    x = _conv3d_as_2d(P, "quant_conv", x)
    return x[:, : cfg.z_dim].unsqueeze(2)


def image_to_latents(P: Params, image: torch.Tensor, cfg: VaeConfig = VaeConfig()) -> torch.Tensor:
    """_encode_vae_image (pipeline_qwen_image_edit.py:459-480): (mean - latents_mean) / latents_std, then packed."""
    z = vae_encode(P, image, cfg)
    mean = torch.tensor(cfg.latents_mean).view(1, cfg.z_dim, 1, 1, 1).to(z)
    std = torch.tensor(cfg.latents_std).view(1, cfg.z_dim, 1, 1, 1).to(z)
    return pack_latents(((z - mean) / std)[:, :, 0])


def vae_encoder_param_shapes(cfg: VaeConfig = VaeConfig()) -> dict:
    """Encoder-side parameter names/shapes (time_conv weights of the 3-D downsamplers exist in checkpoints, unused for T=1)."""
    s: dict[str, tuple] = {}
    dims = [cfg.base_dim * u for u in [1] + list(cfg.dim_mult)]

    def conv(name, i, o, k):
        s[name + ".weight"] = (o, i, k, k, k)
        s[name + ".bias"] = (o,)

    def res(name, i, o):
        s[name + ".norm1.gamma"] = (i, 1, 1, 1)
        conv(name + ".conv1", i, o, 3)
        s[name + ".norm2.gamma"] = (o, 1, 1, 1)
        conv(name + ".conv2", o, o, 3)
        if i != o:
            conv(name + ".conv_shortcut", i, o, 1)

    conv("encoder.conv_in", 3, dims[0], 3)
    k = 0
    for i, (i_dim, o_dim) in enumerate(zip(dims[:-1], dims[1:])):
        cur = i_dim
        for _ in range(cfg.num_res_blocks):
            res(f"encoder.down_blocks.{k}", cur, o_dim)
            cur = o_dim
            k += 1
        if i != len(cfg.dim_mult) - 1:
            s[f"encoder.down_blocks.{k}.resample.1.weight"] = (o_dim, o_dim, 3, 3)
            s[f"encoder.down_blocks.{k}.resample.1.bias"] = (o_dim,)
            k += 1
    top = dims[-1]
    res("encoder.mid_block.resnets.0", top, top)
    s["encoder.mid_block.attentions.0.norm.gamma"] = (top, 1, 1)
    s["encoder.mid_block.attentions.0.to_qkv.weight"] = (top * 3, top, 1, 1)
    s["encoder.mid_block.attentions.0.to_qkv.bias"] = (top * 3,)
    s["encoder.mid_block.attentions.0.proj.weight"] = (top, top, 1, 1)
    s["encoder.mid_block.attentions.0.proj.bias"] = (top,)
    res("encoder.mid_block.resnets.1", top, top)
    s["encoder.norm_out.gamma"] = (top, 1, 1, 1)
    conv("encoder.conv_out", top, cfg.z_dim * 2, 3)
    conv("quant_conv", cfg.z_dim * 2, cfg.z_dim * 2, 1)
    return s


def make_vae_encoder_params(seed: int = 8765, cfg: VaeConfig = VaeConfig(), dtype=torch.float32) -> Params:
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for name, shape in vae_encoder_param_shapes(cfg).items():
        if name.endswith(".weight"):
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            if len(shape) == 5:
                fan_in = fan_in // shape[2]
            t = torch.randn(shape, generator=g, dtype=torch.float32) / math.sqrt(fan_in)
        elif name.endswith(".gamma"):
            t = torch.ones(shape)
        else:
            t = torch.zeros(shape)
        P[name] = t.to(dtype)
    return P


def latents_to_vae_input(latents: torch.Tensor, height: int, width: int, cfg: VaeConfig = VaeConfig()) -> torch.Tensor:
    """pipeline_qwen_image.py:736-746: unpack, then latents / (1/std) + mean."""
    z = unpack_latents(latents, height, width)
    mean = torch.tensor(cfg.latents_mean).view(1, cfg.z_dim, 1, 1, 1).to(z.dtype)
    inv_std = 1.0 / torch.tensor(cfg.latents_std).view(1, cfg.z_dim, 1, 1, 1).to(z.dtype)
    return z / inv_std + mean


# =============================================================================== synthetic parameters
def dit_param_shapes(num_layers: int, num_heads: int = 24, head_dim: int = 128, joint_dim: int = 3584,
                     in_channels: int = 64, out_channels: int = 16, patch: int = 2) -> dict:
    """Parameter names/shapes of QwenImageTransformer2DModel (qwen_image_transformer.py:651-690), in
    `named_parameters()` order so that seeded generation is reproducible everywhere."""
    D = num_heads * head_dim
    s: dict[str, tuple] = {}
    s["time_text_embed.timestep_embedder.linear_1.weight"] = (D, 256)
    s["time_text_embed.timestep_embedder.linear_1.bias"] = (D,)
    s["time_text_embed.timestep_embedder.linear_2.weight"] = (D, D)
    s["time_text_embed.timestep_embedder.linear_2.bias"] = (D,)
    s["txt_norm.weight"] = (joint_dim,)
    s["img_in.weight"] = (D, in_channels)
    s["img_in.bias"] = (D,)
    s["txt_in.weight"] = (D, joint_dim)
    s["txt_in.bias"] = (D,)
    for i in range(num_layers):
        p = f"transformer_blocks.{i}"
        s[p + ".img_mod.1.weight"] = (6 * D, D)
        s[p + ".img_mod.1.bias"] = (6 * D,)
        s[p + ".attn.to_qkv.weight"] = (3 * D, D)
        s[p + ".attn.to_qkv.bias"] = (3 * D,)
        s[p + ".attn.norm_q.weight"] = (head_dim,)
        s[p + ".attn.norm_k.weight"] = (head_dim,)
        s[p + ".attn.add_kv_proj.weight"] = (3 * D, D)
        s[p + ".attn.add_kv_proj.bias"] = (3 * D,)
        s[p + ".attn.to_add_out.weight"] = (D, D)
        s[p + ".attn.to_add_out.bias"] = (D,)
        s[p + ".attn.to_out.0.weight"] = (D, D)
        s[p + ".attn.to_out.0.bias"] = (D,)
        s[p + ".attn.norm_added_q.weight"] = (head_dim,)
        s[p + ".attn.norm_added_k.weight"] = (head_dim,)
        s[p + ".img_mlp.net.0.proj.weight"] = (4 * D, D)
        s[p + ".img_mlp.net.0.proj.bias"] = (4 * D,)
        s[p + ".img_mlp.net.2.weight"] = (D, 4 * D)
        s[p + ".img_mlp.net.2.bias"] = (D,)
        s[p + ".txt_mod.1.weight"] = (6 * D, D)
        s[p + ".txt_mod.1.bias"] = (6 * D,)
        s[p + ".txt_mlp.net.0.proj.weight"] = (4 * D, D)
        s[p + ".txt_mlp.net.0.proj.bias"] = (4 * D,)
        s[p + ".txt_mlp.net.2.weight"] = (D, 4 * D)
        s[p + ".txt_mlp.net.2.bias"] = (D,)
    s["norm_out.linear.weight"] = (2 * D, D)
    s["norm_out.linear.bias"] = (2 * D,)
    s["proj_out.weight"] = (patch * patch * out_channels, D)
    s["proj_out.bias"] = (patch * patch * out_channels,)
    return s


def make_dit_params(num_layers: int, seed: int = 1234, std: float = 0.02, bias_std: float = 0.0,
                    norm_jitter: float = 0.0, dtype=torch.float32, **shape_kw) -> Params:
    """Seeded synthetic weights: >=2-D params N(0, std^2), biases N(0, bias_std^2)
    (0 by default), norm weights 1 (+ jitter).  One CPU generator, parameters drawn in
    `dit_param_shapes` order, always in fp32 then cast — bit-identical on every machine."""
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for name, shape in dit_param_shapes(num_layers, **shape_kw).items():
        if len(shape) >= 2:
            t = torch.randn(shape, generator=g, dtype=torch.float32) * std
        elif "norm" in name:
            t = torch.ones(shape) + (torch.randn(shape, generator=g) * norm_jitter if norm_jitter else 0)
        else:
            t = torch.randn(shape, generator=g, dtype=torch.float32) * bias_std if bias_std else torch.zeros(shape)
        P[name] = t.to(dtype)
    return P


def vae_decoder_param_shapes(cfg: VaeConfig = VaeConfig()) -> dict:
    """Decoder-side parameter names/shapes of AutoencoderKLQwenImage (3-D conv weights [O,I,kt,kh,kw])."""
    s: dict[str, tuple] = {}
    dims = cfg.decoder_dims()

    def conv(name, i, o, k):
        s[name + ".weight"] = (o, i, k, k, k)
        s[name + ".bias"] = (o,)

    def res(name, i, o):
        s[name + ".norm1.gamma"] = (i, 1, 1, 1)
        conv(name + ".conv1", i, o, 3)
        s[name + ".norm2.gamma"] = (o, 1, 1, 1)
        conv(name + ".conv2", o, o, 3)
        if i != o:
            conv(name + ".conv_shortcut", i, o, 1)

    conv("post_quant_conv", cfg.z_dim, cfg.z_dim, 1)
    conv("decoder.conv_in", cfg.z_dim, dims[0], 3)
    res("decoder.mid_block.resnets.0", dims[0], dims[0])
    s["decoder.mid_block.attentions.0.norm.gamma"] = (dims[0], 1, 1)
    s["decoder.mid_block.attentions.0.to_qkv.weight"] = (dims[0] * 3, dims[0], 1, 1)
    s["decoder.mid_block.attentions.0.to_qkv.bias"] = (dims[0] * 3,)
    s["decoder.mid_block.attentions.0.proj.weight"] = (dims[0], dims[0], 1, 1)
    s["decoder.mid_block.attentions.0.proj.bias"] = (dims[0],)
    res("decoder.mid_block.resnets.1", dims[0], dims[0])
    n_up = len(cfg.dim_mult)
    for i, (i_dim, o_dim) in enumerate(zip(dims[:-1], dims[1:])):
        if i > 0:
            i_dim = i_dim // 2
        cur = i_dim
        for j in range(cfg.num_res_blocks + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", cur, o_dim)
            cur = o_dim
        if i != n_up - 1:
            s[f"decoder.up_blocks.{i}.upsamplers.0.resample.1.weight"] = (o_dim // 2, o_dim, 3, 3)
            s[f"decoder.up_blocks.{i}.upsamplers.0.resample.1.bias"] = (o_dim // 2,)
            if cfg.temporal_upsample[i]:  # present in the checkpoint, unused for T=1
                conv(f"decoder.up_blocks.{i}.upsamplers.0.time_conv", o_dim, o_dim * 2, 3)
                s[f"decoder.up_blocks.{i}.upsamplers.0.time_conv.weight"] = (o_dim * 2, o_dim, 3, 1, 1)
    s["decoder.norm_out.gamma"] = (dims[-1], 1, 1, 1)
    conv("decoder.conv_out", dims[-1], 3, 3)
    return s


def make_vae_params(seed: int = 4321, cfg: VaeConfig = VaeConfig(), dtype=torch.float32) -> Params:
    """Seeded synthetic decoder weights: conv weights N(0, 1/fan_in) (keeps activations O(1)), biases 0, gammas 1."""
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for name, shape in vae_decoder_param_shapes(cfg).items():
        if name.endswith(".weight"):
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            # only the last temporal slice acts for T=1: scale by the 2-D fan-in
            if len(shape) == 5:
                fan_in = fan_in // shape[2]
            t = torch.randn(shape, generator=g, dtype=torch.float32) / math.sqrt(fan_in)
        elif name.endswith(".gamma"):
            t = torch.ones(shape)
        else:
            t = torch.zeros(shape)
        P[name] = t.to(dtype)
    return P
