"""AutoencoderKLQwenImage — DECODE side, single frame, on the CDNA4 conv kernels.

Mirror of the decoder half of the vendored VAE (reference
vllm_omni/diffusion/models/qwen_image/autoencoder_kl_qwenimage.py:549-664 QwenImageDecoder3d, :839-863 _decode;
the T2I pipeline imports the diffusers original with the same arithmetic).  Parameter names and shapes are the
checkpoint's ([O, I, kt, kh, kw] causal-Conv3d weights, `gamma` norms, `resample.1` upsample convs), so a
diffusers state dict loads unchanged; `load_weights` then derives the kernels' layout once:

  * for one frame the causal temporal padding puts two ZERO frames in front (:69-84), so only temporal slice
    [-1] of every 3x3x3 kernel touches data -> each conv is a 2-D conv with weight[:, :, -1] (1/3 of the MACs the
    reference executes);
  * activations run NHWC bf16; weights are packed [O, kh, kw, I];
  * nearest-exact x2 upsample (:112-124, fp32 round trip in the reference) is fused into the following conv's
    gather (source = dst >> 1), the residual add into the conv epilogue, the final clamp(-1, 1) into conv_out;
  * the first-chunk "Rep" path skips `time_conv` (:170-176), so upsample3d == upsample2d here.
  * the single-head mid-block attention (:305-330, 16384 tokens x 384 ch at 1024^2) runs as
    GEMM(q k^T) -> row softmax -> GEMM(P v) on the MFMA GEMM kernel.
"""
from __future__ import annotations

import math
from collections.abc import Iterable

import torch
import torch.nn as nn

from .... import ops

BF16 = torch.bfloat16

LATENTS_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
                0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
LATENTS_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
               3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


class _VaeConfig:
    def __init__(self, base_dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2,
                 temperal_downsample=(False, True, True)):
        self.base_dim, self.z_dim, self.dim_mult, self.num_res_blocks = base_dim, z_dim, tuple(dim_mult), num_res_blocks
        self.temporal_upsample = tuple(temperal_downsample[::-1])
        self.latents_mean, self.latents_std = list(LATENTS_MEAN), list(LATENTS_STD)


def decoder_param_shapes(cfg: _VaeConfig) -> dict[str, tuple]:
    """Decoder-side checkpoint names -> shapes (the subset of the state dict this module owns)."""
    s: dict[str, tuple] = {}
    dims = [cfg.base_dim * u for u in [cfg.dim_mult[-1]] + list(cfg.dim_mult[::-1])]

    def conv(n, i, o, k):
        s[n + ".weight"], s[n + ".bias"] = (o, i, k, k, k), (o,)

    def res(n, i, o):
        s[n + ".norm1.gamma"] = (i, 1, 1, 1)
        conv(n + ".conv1", i, o, 3)
        s[n + ".norm2.gamma"] = (o, 1, 1, 1)
        conv(n + ".conv2", o, o, 3)
        if i != o:
            conv(n + ".conv_shortcut", i, o, 1)

    conv("post_quant_conv", cfg.z_dim, cfg.z_dim, 1)
    conv("decoder.conv_in", cfg.z_dim, dims[0], 3)
    res("decoder.mid_block.resnets.0", dims[0], dims[0])
    a = "decoder.mid_block.attentions.0"
    s[a + ".norm.gamma"] = (dims[0], 1, 1)
    s[a + ".to_qkv.weight"], s[a + ".to_qkv.bias"] = (dims[0] * 3, dims[0], 1, 1), (dims[0] * 3,)
    s[a + ".proj.weight"], s[a + ".proj.bias"] = (dims[0], dims[0], 1, 1), (dims[0],)
    res("decoder.mid_block.resnets.1", dims[0], dims[0])
    for i, (i_dim, o_dim) in enumerate(zip(dims[:-1], dims[1:])):
        if i > 0:
            i_dim //= 2
        cur = i_dim
        for j in range(cfg.num_res_blocks + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", cur, o_dim)
            cur = o_dim
        if i != len(cfg.dim_mult) - 1:
            u = f"decoder.up_blocks.{i}.upsamplers.0"
            s[u + ".resample.1.weight"], s[u + ".resample.1.bias"] = (o_dim // 2, o_dim, 3, 3), (o_dim // 2,)
    s["decoder.norm_out.gamma"] = (dims[-1], 1, 1, 1)
    conv("decoder.conv_out", dims[-1], 3, 3)
    return s


def encoder_param_shapes(cfg: _VaeConfig) -> dict[str, tuple]:
    """Encoder-side checkpoint names -> shapes (QwenImageEncoder3d :372-477 + quant_conv; the time_conv weights of the 3-D
    downsamplers exist in checkpoints but only act on later video chunks, :201-211)."""
    s: dict[str, tuple] = {}
    dims = [cfg.base_dim * u for u in [1] + list(cfg.dim_mult)]

    def conv(n, i, o, k):
        s[n + ".weight"], s[n + ".bias"] = (o, i, k, k, k), (o,)

    def res(n, i, o):
        s[n + ".norm1.gamma"] = (i, 1, 1, 1)
        conv(n + ".conv1", i, o, 3)
        s[n + ".norm2.gamma"] = (o, 1, 1, 1)
        conv(n + ".conv2", o, o, 3)
        if i != o:
            conv(n + ".conv_shortcut", i, o, 1)

    conv("encoder.conv_in", 3, dims[0], 3)
    k = 0
    for i, (i_dim, o_dim) in enumerate(zip(dims[:-1], dims[1:])):
        cur = i_dim
        for _ in range(cfg.num_res_blocks):
            res(f"encoder.down_blocks.{k}", cur, o_dim)
            cur, k = o_dim, k + 1
        if i != len(cfg.dim_mult) - 1:
            s[f"encoder.down_blocks.{k}.resample.1.weight"], s[f"encoder.down_blocks.{k}.resample.1.bias"] = (o_dim, o_dim, 3, 3), (o_dim,)
            k += 1
    top = dims[-1]
    res("encoder.mid_block.resnets.0", top, top)
    a = "encoder.mid_block.attentions.0"
    s[a + ".norm.gamma"] = (top, 1, 1)
    s[a + ".to_qkv.weight"], s[a + ".to_qkv.bias"] = (top * 3, top, 1, 1), (top * 3,)
    s[a + ".proj.weight"], s[a + ".proj.bias"] = (top, top, 1, 1), (top,)
    res("encoder.mid_block.resnets.1", top, top)
    s["encoder.norm_out.gamma"] = (top, 1, 1, 1)
    conv("encoder.conv_out", top, cfg.z_dim * 2, 3)
    conv("quant_conv", cfg.z_dim * 2, cfg.z_dim * 2, 1)
    return s


class AutoencoderKLQwenImage(nn.Module):
    """`decode(z)`: de-normalised latents [B, 16, 1, h, w] -> image [B, 3, 1, 8h, 8w].  With `with_encoder=True` also
    `encode(image)` [B, 3, 1, H, W] -> posterior mean [B, 16, 1, H/8, W/8] (what the Edit pipelines feed the DiT)."""

    def __init__(self, device=None, dtype=BF16, with_encoder: bool = False, **cfg_kw):
        super().__init__()
        self.config = _VaeConfig(**cfg_kw)
        self.dtype_ = dtype
        dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self._shapes = decoder_param_shapes(self.config)
        self.with_encoder = with_encoder
        if with_encoder:
            self._shapes.update(encoder_param_shapes(self.config))
        self.params = nn.ParameterDict()
        self._names = {}
        for name, shape in self._shapes.items():
            key = name.replace(".", "__")
            self._names[name] = key
            self.params[key] = nn.Parameter(torch.empty(shape, device=dev, dtype=dtype), requires_grad=False)
        self._packed: dict[str, torch.Tensor] | None = None
        # memory savers of the reference's VAE (autoencoder_kl_qwenimage.py:713-730; set from od_config.vae_use_slicing /
        # vae_use_tiling by the registry, registry.py:88-92)
        self.use_slicing = False
        self.use_tiling = False
        self.tile_sample_min_height = self.tile_sample_min_width = 256
        self.tile_sample_stride_height = self.tile_sample_stride_width = 192
        self.spatial_compression_ratio = 8

    # ---- reference switches (:742-773) ------------------------------------------------------------------------------------------
    def enable_tiling(self, tile_sample_min_height=None, tile_sample_min_width=None, tile_sample_stride_height=None,
                      tile_sample_stride_width=None) -> None:
        self.use_tiling = True
        self.tile_sample_min_height = tile_sample_min_height or self.tile_sample_min_height
        self.tile_sample_min_width = tile_sample_min_width or self.tile_sample_min_width
        self.tile_sample_stride_height = tile_sample_stride_height or self.tile_sample_stride_height
        self.tile_sample_stride_width = tile_sample_stride_width or self.tile_sample_stride_width

    def disable_tiling(self) -> None:
        self.use_tiling = False

    def enable_slicing(self) -> None:
        self.use_slicing = True

    def disable_slicing(self) -> None:
        self.use_slicing = False

    @property
    def dtype(self):
        return self.dtype_

    @property
    def device(self):
        return next(iter(self.params.values())).device

    def state_names(self) -> list[str]:
        return list(self._shapes.keys())

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        loaded = set()
        for name, w in weights:
            if name not in self._names:
                continue  # time_conv (and, without with_encoder, encoder / quant_conv): not on this build's path
            p = self.params[self._names[name]]
            if p.shape != w.shape:
                raise ValueError(f"{name}: expected {tuple(p.shape)}, got {tuple(w.shape)}")
            p.data.copy_(w)
            loaded.add(name)
        self._packed = None
        return loaded

    def init_random_(self, seed: int = 4321):
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, shape in self._shapes.items():
            p = self.params[self._names[name]]
            if name.endswith(".weight"):
                fan = math.prod(shape[1:]) // (shape[2] if len(shape) == 5 else 1)
                p.data.copy_(torch.randn(shape, device=self.device, generator=g) / math.sqrt(fan))
            elif name.endswith(".gamma"):
                p.data.fill_(1.0)
            else:
                p.data.zero_()
        self._packed = None
        return self

    # ------------------------------------------------------------------ derived kernel layout
    def _pack(self) -> dict[str, torch.Tensor]:
        if self._packed is not None:
            return self._packed
        out: dict[str, torch.Tensor] = {}
        for name in self._shapes:
            p = self.params[self._names[name]].data
            if name.endswith(".weight"):
                if p.dim() == 5:
                    p = p[:, :, -1]                       # only the last temporal slice sees the single frame
                p = p.permute(0, 2, 3, 1)                         # [O, kh, kw, I]
                if p.shape[-1] % 8:                                 # encoder.conv_in: 3 input channels -> zero-padded to 8
                    p = torch.nn.functional.pad(p, (0, 8 - p.shape[-1] % 8))
                out[name] = p.contiguous()
            else:
                out[name] = p.reshape(-1).contiguous()
        self._packed = out
        return out

    ATTN_Q_CHUNK = 8192
    flash_mid_attention = True        # False: the GEMM -> softmax -> GEMM path (any channel count; the flash kernel's cross-check)

    # ------------------------------------------------------------------ blocks (NHWC bf16)
    def _res_block(self, W, pre, x):
        h = ops.vae_conv2d(x, W[pre + ".conv_shortcut.weight"], W[pre + ".conv_shortcut.bias"]) \
            if (pre + ".conv_shortcut.weight") in W else x
        y = ops.vae_rmsnorm_silu(x, W[pre + ".norm1.gamma"])
        y = ops.vae_conv2d(y, W[pre + ".conv1.weight"], W[pre + ".conv1.bias"])
        y = ops.vae_rmsnorm_silu(y, W[pre + ".norm2.gamma"])
        return ops.vae_conv2d(y, W[pre + ".conv2.weight"], W[pre + ".conv2.bias"], res=h)

    def _attn_block(self, W, pre, x):
        B, H, Wd, Cc = x.shape
        tok = H * Wd
        if Cc % 64:
            raise NotImplementedError("mid-block attention needs the channel count to be a multiple of 64")
        # the P.V product contracts over the keys: its K dimension must be a multiple of 64 for the GEMM kernels.  Token
        # counts that are not (e.g. a 352 x 352 image: 44 x 44 latent) get masked pad keys: score -inf, V^T column 0.
        tokp = (tok + 63) // 64 * 64
        xn = ops.vae_rmsnorm_silu(x, W[pre + ".norm.gamma"], silu=False)
        wqkv = W[pre + ".to_qkv.weight"].reshape(3 * Cc, Cc)
        bqkv = W[pre + ".to_qkv.bias"]
        if Cc == ops.VAE_ATTENTION_CHANNELS and self.flash_mid_attention:
            # one fused q / k / v projection over all images, then the flash kernel on its column slices: the tok x tok score
            # matrix (1 GiB per 1024^2 image through HBM four times on the path below) never leaves the registers
            qkv = ops.linear(xn.reshape(B * tok, Cc), wqkv, bqkv).view(B, tok, 3 * Cc)
            o = ops.vae_attention(qkv[:, :, :Cc], qkv[:, :, Cc:2 * Cc], qkv[:, :, 2 * Cc:], 1.0 / math.sqrt(Cc)).view(B, H, Wd, Cc)
            return ops.vae_conv2d(o, W[pre + ".proj.weight"], W[pre + ".proj.bias"], res=x)
        outs = []
        for b in range(B):
            t = xn[b].reshape(tok, Cc)
            if tokp != tok:                                             # zero pad rows: V^T pad columns come out 0
                t = torch.nn.functional.pad(t, (0, 0, 0, tokp - tok)).contiguous()
            q = ops.linear(t, wqkv[:Cc], bqkv[:Cc])
            k = ops.linear(t, wqkv[Cc:2 * Cc], bqkv[Cc:2 * Cc])
            vt = ops.linear(wqkv[2 * Cc:].contiguous(), t)              # V^T [C, tokp] (bias folded below)
            # query rows in chunks: the score buffer is [chunk, tok] bf16 (256 MiB at 1024^2, 1 GiB at 2048^2) instead
            # of [tok, tok] (512 MiB / 8.6 GB)
            o_b = torch.empty(tok, Cc, dtype=BF16, device=x.device)
            # P V is tall with a long K and few output tiles (8192 x 384 x 16384: 64 workgroups): split-K workspace for it
            ws = torch.empty(8 * min(tok, self.ATTN_Q_CHUNK) * Cc, dtype=torch.float32, device=x.device)
            for r0 in range(0, tok, self.ATTN_Q_CHUNK):
                r1 = min(tok, r0 + self.ATTN_Q_CHUNK)
                s = ops.linear(q[r0:r1], k)                             # [chunk, tokp] scores
                if tokp != tok:
                    s[:, tok:] = float("-inf")                          # pad keys get probability 0
                ops.softmax_rows_(s, 1.0 / math.sqrt(Cc))
                ops.gemm([ops.GemmGroupArgs(s, vt, bqkv[2 * Cc:], o_b[r0:r1])], splitk_ws=ws,
                         kernel_hint=ops.GEMM_KERNEL_SPLITK_TALL)        # P V + b_v  (rows of P sum to 1)
            outs.append(o_b)
        o = torch.stack(outs).view(B, H, Wd, Cc)
        return ops.vae_conv2d(o, W[pre + ".proj.weight"], W[pre + ".proj.bias"], res=x)

    @torch.no_grad()
    def encode(self, image: torch.Tensor) -> torch.Tensor:
        """image [B, 3, 1, H, W] in [-1, 1] -> posterior MEAN [B, z_dim, 1, H/8, W/8] (reference encode :811-835 + _encode
        :788-810 for one frame + DiagonalGaussianDistribution.mode(); the Edit pipelines use sample_mode="argmax",
        pipeline_qwen_image_edit.py:459-467).  `use_slicing`: one image at a time (:828-830); `use_tiling`: images larger than
        the tile go through `tiled_encode` (:791-792)."""
        if self.use_slicing and image.shape[0] > 1:
            return torch.cat([self._encode(x) for x in image.split(1)])
        return self._encode(image)

    def _encode(self, image: torch.Tensor) -> torch.Tensor:
        if image.dim() == 5 and self.use_tiling and (image.shape[4] > self.tile_sample_min_width or
                                                     image.shape[3] > self.tile_sample_min_height):
            return self.tiled_encode(image)
        return self._encode_plain(image)

    # ---- spatial tiling (reference tiled_encode :905-969, tiled_decode :971-1031, blend_v / blend_h :889-903) ----------------------
    @staticmethod
    def _blend(a: torch.Tensor, b: torch.Tensor, extent: int, dim: int) -> torch.Tensor:
        """The first `extent` rows (dim = -2) / columns (dim = -1) of b become a cross-fade from the last ones of a, IN PLACE on b:
        b[y] = a[-extent + y] * (1 - y / extent) + b[y] * (y / extent).  The reference runs this loop in the VAE's dtype (bf16)
        with Python-float weights: each product and the sum are rounded to bf16 — reproduced here in one vectorised step."""
        extent = min(a.shape[dim], b.shape[dim], extent)
        if extent <= 0:
            return b
        y = torch.arange(extent, dtype=torch.float64)
        shape = [1] * b.dim()
        shape[dim] = extent
        w_b = (y / extent).to(torch.float32).view(shape).to(b.device)
        w_a = (1 - y / extent).to(torch.float32).view(shape).to(b.device)
        a_part = a.narrow(dim, a.shape[dim] - extent, extent)
        b_part = b.narrow(dim, 0, extent)
        pa = (a_part.float() * w_a).to(b.dtype)
        pb = (b_part.float() * w_b).to(b.dtype)
        b_part.copy_((pa.float() + pb.float()).to(b.dtype))
        return b

    @classmethod
    def _stitch(cls, rows, blend_h: int, blend_w: int, stride_h: int, stride_w: int) -> torch.Tensor:
        """(:955-968, :1014-1028) every tile is blended with the already blended tile above and the one to its left, cropped to
        the stride; rows and columns are concatenated."""
        out_rows = []
        for i, row in enumerate(rows):
            out_row = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = cls._blend(rows[i - 1][j], tile, blend_h, -2)
                if j > 0:
                    tile = cls._blend(row[j - 1], tile, blend_w, -1)
                out_row.append(tile[:, :, :, :stride_h, :stride_w])
            out_rows.append(torch.cat(out_row, dim=-1))
        return torch.cat(out_rows, dim=3)

    TILE_BATCH = 16                                   # equal-sized tiles decoded / encoded per call (a 1024^2 image's worth of pixels)

    def _map_tiles(self, x: torch.Tensor, tmin_h: int, tmin_w: int, str_h: int, str_w: int, fn):
        """Cut x [B, C, 1, H, W] into overlapping tiles, run fn on them — equal-sized tiles batched TILE_BATCH at a time (every
        kernel of the VAE treats batch entries independently) — and return them as rows[i][j] of [B, C', 1, h', w'] tensors."""
        B, H, Wd = x.shape[0], x.shape[3], x.shape[4]
        pos = [(i, j) for i in range(0, H, str_h) for j in range(0, Wd, str_w)]
        tiles = {p: x[:, :, :, p[0]:p[0] + tmin_h, p[1]:p[1] + tmin_w] for p in pos}
        done: dict[tuple, torch.Tensor] = {}
        by_shape: dict[tuple, list] = {}
        for p in pos:
            by_shape.setdefault(tuple(tiles[p].shape[3:]), []).append(p)
        for _shape, ps in by_shape.items():
            for c0 in range(0, len(ps), max(1, self.TILE_BATCH // B)):
                part = ps[c0:c0 + max(1, self.TILE_BATCH // B)]
                out = fn(torch.cat([tiles[p] for p in part]).contiguous())
                for k, p in enumerate(part):
                    done[p] = out[k * B:(k + 1) * B].clone()
        return [[done[(i, j)] for j in range(0, Wd, str_w)] for i in range(0, H, str_h)]

    @torch.no_grad()
    def tiled_encode(self, image: torch.Tensor) -> torch.Tensor:
        """(:905-969) image tiles of 256^2 every 192 pixels through the encoder, latent overlaps of 8 cross-faded -> posterior mean
        [B, z_dim, 1, H/8, W/8].  (The reference blends mean | logvar together; the blend is linear and per channel, so blending
        the mean half alone is the same.)"""
        sr = self.spatial_compression_ratio
        H, Wd = image.shape[3], image.shape[4]
        tl_h, tl_w = self.tile_sample_min_height // sr, self.tile_sample_min_width // sr
        ts_h, ts_w = self.tile_sample_stride_height // sr, self.tile_sample_stride_width // sr
        rows = self._map_tiles(image, self.tile_sample_min_height, self.tile_sample_min_width, self.tile_sample_stride_height,
                               self.tile_sample_stride_width, self._encode_plain)
        return self._stitch(rows, tl_h - ts_h, tl_w - ts_w, ts_h, ts_w)[:, :, :, : H // sr, : Wd // sr].contiguous()

    @torch.no_grad()
    def tiled_decode(self, z: torch.Tensor, return_dict: bool = False):
        """(:971-1031) latent tiles of 32^2 every 24 positions decoded independently, pixel overlaps of 64 cross-faded, every tile
        contributes its first 192 x 192 pixels.  As in the reference the stitched image is NOT clamped (`_decode` leaves for
        `tiled_decode` before its clamp, :844-845 vs :857): values outside [-1, 1] reach the image processor."""
        sr = self.spatial_compression_ratio
        h, w = z.shape[3], z.shape[4]
        tl_h, tl_w = self.tile_sample_min_height // sr, self.tile_sample_min_width // sr
        ts_h, ts_w = self.tile_sample_stride_height // sr, self.tile_sample_stride_width // sr
        rows = self._map_tiles(z, tl_h, tl_w, ts_h, ts_w, lambda t: self._decode_plain_or_bordered(t, clamp=None))
        img = self._stitch(rows, self.tile_sample_min_height - self.tile_sample_stride_height,
                           self.tile_sample_min_width - self.tile_sample_stride_width, self.tile_sample_stride_height,
                           self.tile_sample_stride_width)[:, :, :, : h * sr, : w * sr].contiguous()
        return (img,)

    @torch.no_grad()
    def _encode_plain(self, image: torch.Tensor) -> torch.Tensor:
        if not self.with_encoder:
            raise RuntimeError("this VAE was built without its encoder (with_encoder=True)")
        if image.dim() != 5 or image.shape[2] != 1 or image.shape[1] != 3:
            raise NotImplementedError("single-frame RGB images only")
        if image.shape[3] % 8 or image.shape[4] % 8:
            raise ValueError("image height and width must be multiples of 8")
        W = self._pack()
        c = self.config
        x = image[:, :, 0].permute(0, 2, 3, 1).to(BF16)                      # NHWC
        x = torch.nn.functional.pad(x, (0, 5)).contiguous()                   # 3 -> 8 channels (zeros), matches the packed weight
        x = ops.vae_conv2d(x, W["encoder.conv_in.weight"], W["encoder.conv_in.bias"])
        k = 0
        for i in range(len(c.dim_mult)):
            for _ in range(c.num_res_blocks):
                x = self._res_block(W, f"encoder.down_blocks.{k}", x)
                k += 1
            if i != len(c.dim_mult) - 1:
                d = f"encoder.down_blocks.{k}.resample.1"
                x = ops.vae_conv2d(x, W[d + ".weight"], W[d + ".bias"], downsample2x=True)
                k += 1
        x = self._res_block(W, "encoder.mid_block.resnets.0", x)
        x = self._attn_block(W, "encoder.mid_block.attentions.0", x)
        x = self._res_block(W, "encoder.mid_block.resnets.1", x)
        x = ops.vae_rmsnorm_silu(x, W["encoder.norm_out.gamma"])
        x = ops.vae_conv2d(x, W["encoder.conv_out.weight"], W["encoder.conv_out.bias"])
        x = ops.vae_conv2d(x, W["quant_conv.weight"], W["quant_conv.bias"])
        return x[..., : c.z_dim].permute(0, 3, 1, 2).unsqueeze(2).contiguous()   # mean half, [B, 16, 1, h, w]

    # ---- decode over ZERO-BORDERED rasters -------------------------------------------------------------------------
    # Between conv_in and conv_out every activation is [B, H + 2, W + 2, C] with a resident one-pixel zero border (the
    # reference re-creates it with F.pad in front of every conv, autoencoder_kl_qwenimage.py:80-84): a 3x3 conv is then a
    # GEMM over nine row-shifted views of the same matrix, fed by plain LDS-DMA (vae.hip conv_bordered_kernel).  The conv
    # kernel writes zeros at border positions, the norm maps 0 to 0, the upsample kernel writes its own border.
    # The norm + SiLU in front of a conv is produced by the conv BEFORE it (ops.vae_conv2d norm_gamma: a second output of the
    # same launch where a workgroup holds all channels of a pixel): `xn` is silu(norm1(x)) when the producer of x already made
    # it, `next_gamma` the norm that will consume this block's output.
    def _res_block_b(self, W, pre, x, xn=None, next_gamma=None, keep_raw=True):
        kw = dict(x_bordered=True, y_bordered=True)
        h = ops.vae_conv2d(x, W[pre + ".conv_shortcut.weight"], W[pre + ".conv_shortcut.bias"], **kw) \
            if (pre + ".conv_shortcut.weight") in W else x
        if xn is None:
            xn = ops.vae_rmsnorm_silu(x, W[pre + ".norm1.gamma"])
        _, y = ops.vae_conv2d(xn, W[pre + ".conv1.weight"], W[pre + ".conv1.bias"], norm_gamma=W[pre + ".norm2.gamma"],
                              keep_raw=False, **kw)
        out = ops.vae_conv2d(y, W[pre + ".conv2.weight"], W[pre + ".conv2.bias"], res=h, norm_gamma=next_gamma, keep_raw=keep_raw, **kw)
        return out if next_gamma is not None else (out, None)

    @staticmethod
    def _add_border(x):
        return torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1))

    def _bordered_ok(self) -> bool:
        c = self.config
        return all((c.base_dim * m) % 32 == 0 for m in c.dim_mult)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = False):
        """z [B, z_dim, 1, h, w] -> [B, 3, 1, 8h, 8w] (reference decode :865-887 + _decode :839-863, one frame): clamped to
        [-1, 1] on the plain path; `use_slicing`: one latent at a time (:879-881); `use_tiling`: latents larger than a tile go
        through `tiled_decode` (:844-845)."""
        if z.dim() != 5 or z.shape[2] != 1:
            raise NotImplementedError("single-frame (image) decode only")
        if self.use_slicing and z.shape[0] > 1:
            return (torch.cat([self._decode(zs)[0] for zs in z.split(1)]),)
        return self._decode(z)

    def _decode(self, z: torch.Tensor):
        sr = self.spatial_compression_ratio
        if self.use_tiling and (z.shape[4] > self.tile_sample_min_width // sr or z.shape[3] > self.tile_sample_min_height // sr):
            return self.tiled_decode(z)
        return (self._decode_plain_or_bordered(z, clamp=(-1.0, 1.0)),)

    @torch.no_grad()
    def _decode_plain_or_bordered(self, z: torch.Tensor, clamp=(-1.0, 1.0)) -> torch.Tensor:
        if not self._bordered_ok():
            return self._decode_plain(z, clamp)[0]
        W = self._pack()
        c = self.config
        x = z[:, :, 0].permute(0, 2, 3, 1).contiguous().to(BF16)      # NHWC
        x = ops.vae_conv2d(x, W["post_quant_conv.weight"], W["post_quant_conv.bias"])
        x = ops.vae_conv2d(x, W["decoder.conv_in.weight"], W["decoder.conv_in.bias"])      # Cin = z_dim: the gather kernel
        x = self._add_border(x)
        x, _ = self._res_block_b(W, "decoder.mid_block.resnets.0", x)
        x = self._add_border(self._attn_block(W, "decoder.mid_block.attentions.0", x[:, 1:-1, 1:-1].contiguous()))
        # the chain of residual blocks / upsamplers behind the attention: every conv that feeds a norm is told its gamma
        chain = ["decoder.mid_block.resnets.1"]
        n_up = len(c.dim_mult)
        for i in range(n_up):
            chain += [f"decoder.up_blocks.{i}.resnets.{j}" for j in range(c.num_res_blocks + 1)]
            if i != n_up - 1:
                chain.append(f"decoder.up_blocks.{i}.upsamplers.0.resample.1")
        xn = None
        for k, name in enumerate(chain):
            nxt = chain[k + 1] if k + 1 < len(chain) else None
            if nxt is None:
                g = W["decoder.norm_out.gamma"]
            else:
                g = W[nxt + ".norm1.gamma"] if "resnets" in nxt else None      # an upsampler takes x raw
            if "resnets" in name:
                x, xn = self._res_block_b(W, name, x, xn, next_gamma=g, keep_raw=nxt is not None)
            else:
                r = ops.vae_conv2d(x, W[name + ".weight"], W[name + ".bias"], upsample2x=True, x_bordered=True,
                                   y_bordered=True, norm_gamma=g)      # the x2 upsample happens in the conv's operand fetch
                x, xn = r if g is not None else (r, None)
        x = ops.vae_conv2d(xn, W["decoder.conv_out.weight"], W["decoder.conv_out.bias"], clamp=clamp, x_bordered=True)
        return x.permute(0, 3, 1, 2).unsqueeze(2)                      # [B, 3, 1, H, W]

    @torch.no_grad()
    def _decode_plain(self, z: torch.Tensor, clamp=(-1.0, 1.0)):
        """The same network over plain NHWC rasters (the gather kernel does the zero padding per tap): channel counts the
        bordered kernel is not built for."""
        W = self._pack()
        c = self.config
        x = z[:, :, 0].permute(0, 2, 3, 1).contiguous().to(BF16)      # NHWC
        x = ops.vae_conv2d(x, W["post_quant_conv.weight"], W["post_quant_conv.bias"])
        x = ops.vae_conv2d(x, W["decoder.conv_in.weight"], W["decoder.conv_in.bias"])
        x = self._res_block(W, "decoder.mid_block.resnets.0", x)
        x = self._attn_block(W, "decoder.mid_block.attentions.0", x)
        x = self._res_block(W, "decoder.mid_block.resnets.1", x)
        n_up = len(c.dim_mult)
        for i in range(n_up):
            for j in range(c.num_res_blocks + 1):
                x = self._res_block(W, f"decoder.up_blocks.{i}.resnets.{j}", x)
            if i != n_up - 1:
                u = f"decoder.up_blocks.{i}.upsamplers.0.resample.1"
                x = ops.vae_conv2d(x, W[u + ".weight"], W[u + ".bias"], upsample2x=True)
        x = ops.vae_rmsnorm_silu(x, W["decoder.norm_out.gamma"])
        x = ops.vae_conv2d(x, W["decoder.conv_out.weight"], W["decoder.conv_out.bias"], clamp=clamp)
        img = x.permute(0, 3, 1, 2).unsqueeze(2)                       # [B, 3, 1, H, W]
        return (img,)
