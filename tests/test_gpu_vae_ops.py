"""VAE kernels over zero-bordered rasters (vae.hip conv_bordered_kernel / upsample2x_bordered_kernel) against torch fp32 —
the conv the reference runs as F.pad + Conv3d on the last temporal slice (autoencoder_kl_qwenimage.py:69-84), the upsample of
QwenImageUpsample (:112-124)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rnd(shape, seed, s=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * s).to(torch.bfloat16)


def _border(x):
    return torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1))


@pytest.mark.parametrize("cin,cout,ks,hw,B,use_res", [
    (96, 96, 3, (20, 37), 2, True),        # 512-pixel x 96-channel tiles, ragged last tile, two images
    (192, 384, 3, (33, 18), 1, False),     # 256 x 192 tiles, two channel tiles
    (384, 192, 3, (16, 16), 2, True),
    (192, 192, 3, (40, 24), 1, True),
    (192, 384, 1, (21, 19), 2, False),     # the res-block shortcut: 1x1
    (384, 384, 3, (9, 130), 1, False),     # rows longer than a tile
    (192, 384, 3, (256, 256), 1, True),    # a full round of 512 px x 192 ch tiles: the 128-pixel-per-wave kernel, two runs per tile
    (96, 384, 3, (300, 200), 1, False),    # the same with ragged runs (200 of 256 pixels)
    (96, 96, 3, (256, 1024), 1, True),     # many 256 px x 96 ch tiles, runs shorter than the row
    (192, 192, 1, (512, 512), 1, False),   # 1x1 through the big kernel
])
def test_bordered_conv_matches_torch(cin, cout, ks, hw, B, use_res):
    from vllm_omni_amd import ops

    H, W = hw
    x, w, b = _rnd((B, H, W, cin), 1), _rnd((cout, ks, ks, cin), 2, 0.05), _rnd((cout,), 3)
    res = _rnd((B, H, W, cout), 4) if use_res else None
    ref = torch.nn.functional.conv2d(x.to(DEV).float().permute(0, 3, 1, 2), w.to(DEV).float().permute(0, 3, 1, 2), b.to(DEV).float(),
                                     padding=ks // 2).cpu()                         # fp32 reference (computed on the GPU: size)
    ref = ref.permute(0, 2, 3, 1) + (res.float() if use_res else 0.0)
    y = ops.vae_conv2d(_border(x).to(DEV), w.to(DEV), b.to(DEV), res=_border(res).to(DEV) if use_res else None,
                       x_bordered=True, y_bordered=True).float().cpu()
    assert y.shape == (B, H + 2, W + 2, cout)
    # the border stays exactly zero (the next layer's padding), the interior matches fp32 to bf16 rounding of the output
    assert float(y[:, 0].abs().max()) == 0.0 and float(y[:, -1].abs().max()) == 0.0
    assert float(y[:, :, 0].abs().max()) == 0.0 and float(y[:, :, -1].abs().max()) == 0.0
    got = y[:, 1:-1, 1:-1]
    err = float((got - ref).norm() / ref.norm())
    assert err <= 4e-3, err
    # and the gather kernel (plain rasters) agrees with it
    y_plain = ops.vae_conv2d(x.to(DEV), w.to(DEV), b.to(DEV), res=res.to(DEV) if use_res else None).float().cpu()
    assert float((got - y_plain).norm() / ref.norm()) <= 4e-3


@pytest.mark.parametrize("cin,cout,ks,hw,B,use_res,fused", [
    (96, 96, 3, (20, 37), 2, True, True),        # 256 px x 96 ch tiles: the lane pair of a pixel holds all channels
    (192, 64, 3, (33, 18), 1, False, True),      # fewer channels than the tile: the idle channel runs stay out of the sum
    (96, 96, 3, (256, 1024), 1, True, True),
    (192, 192, 3, (384, 384), 1, True, True),    # 512 px x 192 ch tiles: two waves share a pixel (LDS exchange)
    (96, 192, 1, (512, 300), 1, False, True),    # ... 1x1, ragged runs
    (192, 192, 3, (40, 24), 1, True, False),     # 192 channels on a small raster: 96-channel tiles -> the separate norm pass
    (192, 384, 3, (64, 64), 1, False, False),    # more channels than any tile
])
@pytest.mark.parametrize("silu", [True, False])
def test_conv_with_the_following_norm_as_second_output(cin, cout, ks, hw, B, use_res, fused, silu):
    """ABI v10 norm_gamma: y_norm must be what omni_vae_rmsnorm_silu makes of y (same formula from the bf16-rounded y; only the
    order in which the squares are summed differs), y itself and the zero borders of both unchanged, and with keep_raw=False
    the kernel writes the normed output alone."""
    import ctypes

    from vllm_omni_amd import _native as N
    from vllm_omni_amd import ops

    H, W = hw
    x, w, b = _border(_rnd((B, H, W, cin), 1)).to(DEV), _rnd((cout, ks, ks, cin), 2, 0.05).to(DEV), _rnd((cout,), 3).to(DEV)
    res = _border(_rnd((B, H, W, cout), 4)).to(DEV) if use_res else None
    g = (1.0 + 0.2 * _rnd((cout,), 5).float()).to(torch.bfloat16).to(DEV)
    kw = dict(res=res, x_bordered=True, y_bordered=True)
    y0 = ops.vae_conv2d(x, w, b, **kw)
    n0 = ops.vae_rmsnorm_silu(y0, g, silu=silu)
    y1, n1 = ops.vae_conv2d(x, w, b, norm_gamma=g, norm_silu=silu, **kw)
    assert torch.equal(y0, y1)
    d = (n1.float() - n0.float()).abs()
    ulp = n0.float().abs() * 2.0 ** -7 + 1e-30
    assert float((d / ulp).max()) <= 1.0 and float((d > 0).float().mean()) <= 1e-2     # <= 1 bf16 ulp, on <= 1 % of the values
    for t in (n1,):
        assert float(t[:, 0].abs().max()) == 0.0 and float(t[:, -1].abs().max()) == 0.0
        assert float(t[:, :, 0].abs().max()) == 0.0 and float(t[:, :, -1].abs().max()) == 0.0
    y2, n2 = ops.vae_conv2d(x, w, b, norm_gamma=g, norm_silu=silu, keep_raw=False, **kw)
    assert torch.equal(n2, n1)
    assert (y2 is None) == fused                          # the launcher's promise (omni_vae_conv2d_fuses_norm) and what ops did with it
    p = N.ConvParams()
    p.B, p.Hin, p.Win, p.Cin, p.Cout, p.ksize, p.x_padded, p.y_padded = B, H, W, cin, cout, ks, 1, 1
    p.norm_gamma = g.data_ptr()
    assert bool(N.lib().omni_vae_conv2d_fuses_norm(ctypes.byref(p))) == fused


@pytest.mark.parametrize("B,tok", [(2, 1024), (1, 1936), (3, 200), (1, 16384)])
def test_vae_attention_matches_torch(B, tok):
    """omni_vae_attention (one head of 384 channels, QwenImageAttentionBlock.forward :305-330) against fp32 attention computed by
    torch on the GPU from the same bf16 inputs; q, k, v are column slices of one fused projection (row stride 3 C), token counts
    include ragged tails (1936 = 44 x 44, 200)."""
    import math

    from vllm_omni_amd import ops

    Cc = 384
    qkv = _rnd((B, tok, 3 * Cc), 31, 1.0).to(DEV)
    qkv[..., :Cc] *= 2.0                                               # sharper rows than unit-variance scores
    q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]
    got = ops.vae_attention(q, k, v, 1.0 / math.sqrt(Cc)).float()
    ref = torch.empty_like(got)
    for b in range(B):
        for r0 in range(0, tok, 4096):
            s = (q[b, r0:r0 + 4096].float() @ k[b].float().T) / math.sqrt(Cc)
            ref[b, r0:r0 + 4096] = torch.softmax(s, dim=-1) @ v[b].float()
    torch.cuda.synchronize()
    err = float((got - ref).norm() / ref.norm())
    assert torch.isfinite(got).all() and err <= 8e-3, err              # bf16 P and bf16 output rounding


def test_vae_mid_attention_flash_equals_the_gemm_softmax_gemm_path():
    from vllm_omni_amd.diffusion.models.qwen_image.autoencoder_kl_qwenimage import AutoencoderKLQwenImage

    vae = AutoencoderKLQwenImage(device=DEV)
    vae.init_random_(seed=3)
    W = vae._pack()
    x = _rnd((2, 24, 20, 384), 41, 1.0).to(DEV)
    a = vae._attn_block(W, "decoder.mid_block.attentions.0", x).float()
    vae.flash_mid_attention = False
    b = vae._attn_block(W, "decoder.mid_block.attentions.0", x).float()
    assert float((a - b).norm() / b.norm()) <= 8e-3


@pytest.mark.parametrize("rows,cols,pad", [(64, 16384, 0), (7, 3000, 0), (5, 1936 + 48, 48), (4, 65536, 0), (3, 70000, 0)])
def test_softmax_rows_matches_torch(rows, cols, pad):
    """omni_softmax_rows (the VAE mid-block attention's softmax, autoencoder_kl_qwenimage.py:319 through GEMM -> this -> GEMM):
    the register-resident kernels (<= 16384 and <= 65536 columns) and the three-pass one, with masked (-inf) pad keys."""
    from vllm_omni_amd import ops

    s = _rnd((rows, cols), 11, 6.0)
    if pad:
        s[:, cols - pad:] = float("-inf")
    scale = 0.051
    ref = torch.softmax(s.float() * scale, dim=-1)
    got = ops.softmax_rows_(s.to(DEV).clone(), scale).float().cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 2.0 ** -8 * float(ref.max()) + 1e-6          # bf16 rounding of the probabilities
    assert float((got.sum(-1) - 1).abs().max()) <= 2e-2
    if pad:
        assert float(got[:, cols - pad:].abs().max()) == 0.0


@pytest.mark.parametrize("B,hw,cin,cout,clamp", [
    (2, (19, 23), 96, 3, (-1.0, 1.0)),      # the decoder's conv_out at a ragged size
    (1, (40, 300), 96, 3, (-1.0, 1.0)),     # rows longer than a workgroup's 256 pixels
    (1, (8, 70), 64, 16, None),             # all 16 MFMA columns live
    (1, (5, 33), 128, 4, None),
    (1, (6, 20), 160, 3, None),             # more input channels than conv_few_kernel holds weights for: the gather kernel
])
def test_conv_to_few_channels_reads_bordered_input(B, hw, cin, cout, clamp):
    """conv_out of the decoder (autoencoder_kl_qwenimage.py:737-739): bordered raster in, plain raster out, <= 16 output
    channels — vae.hip conv_few_kernel; against torch fp32 and against the gather kernel over the plain raster."""
    from vllm_omni_amd import ops

    H, W = hw
    x, w, b = _rnd((B, H, W, cin), 5), _rnd((cout, 3, 3, cin), 6, 0.05), _rnd((cout,), 7)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1).permute(0, 2, 3, 1)
    if clamp:
        ref = ref.clamp(*clamp)
    plain = ops.vae_conv2d(x.to(DEV), w.to(DEV), b.to(DEV), clamp=clamp).float().cpu()
    got = ops.vae_conv2d(_border(x).to(DEV), w.to(DEV), b.to(DEV), clamp=clamp, x_bordered=True).float().cpu()
    assert got.shape == ref.shape == plain.shape
    assert float((got - ref).norm() / ref.norm()) <= 4e-3
    assert float((got - plain).norm() / ref.norm()) <= 4e-3


@pytest.mark.parametrize("cin,cout,hw,B,with_norm", [
    (384, 192, (16, 16), 2, False),      # 96-channel tiles
    (96, 96, (37, 21), 1, True),         # odd source sizes, fused norm (lane pair)
    (192, 96, (130, 260), 1, True),      # runs shorter than the row
    (384, 192, (200, 192), 1, True),     # 192-channel tiles (two waves share a pixel), fused norm
    (192, 384, (64, 300), 2, False),     # two channel blocks per tile
])
def test_conv_over_the_upsampled_raster_without_the_upsampled_tensor(cin, cout, hw, B, with_norm):
    """upsample2x on bordered rasters: the conv's operand fetch maps every pixel of the x2 raster to its source pixel, so the
    result must equal the conv over the explicitly upsampled raster (omni_vae_upsample2x_bordered) bit for bit — same operands,
    same order of accumulation."""
    from vllm_omni_amd import ops

    H, W = hw
    x, w, b = _border(_rnd((B, H, W, cin), 21)).to(DEV), _rnd((cout, 3, 3, cin), 22, 0.05).to(DEV), _rnd((cout,), 23).to(DEV)
    g = (1.0 + 0.2 * _rnd((cout,), 24).float()).to(torch.bfloat16).to(DEV) if with_norm else None
    kw = dict(x_bordered=True, y_bordered=True, norm_gamma=g)
    want = ops.vae_conv2d(ops.vae_upsample2x_bordered(x), w, b, **kw)
    got = ops.vae_conv2d(x, w, b, upsample2x=True, **kw)
    if with_norm:
        assert got[0].shape == (B, 2 * H + 2, 2 * W + 2, cout)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    else:
        assert got.shape == (B, 2 * H + 2, 2 * W + 2, cout) and torch.equal(got, want)


def test_bordered_upsample_is_nearest_exact():
    from vllm_omni_amd import ops

    x = _rnd((2, 7, 11, 96), 8)
    ref = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest-exact").permute(0, 2, 3, 1)
    y = ops.vae_upsample2x_bordered(_border(x).to(DEV)).float().cpu()
    assert y.shape == (2, 16, 24, 96)
    assert torch.equal(y[:, 1:-1, 1:-1], ref)
    assert float(y[:, 0].abs().max()) == 0.0 and float(y[:, -1].abs().max()) == 0.0 and float(y[:, :, 0].abs().max()) == 0.0 \
        and float(y[:, :, -1].abs().max()) == 0.0


def test_bordered_conv_rejects_what_it_is_not_built_for():
    from vllm_omni_amd import ops
    from vllm_omni_amd._native import OmniNativeError

    x, w = _rnd((1, 10, 10, 16), 9).to(DEV), _rnd((96, 3, 3, 16), 10).to(DEV)
    with pytest.raises(OmniNativeError):                                   # Cin % 32
        ops.vae_conv2d(x, w, x_bordered=True, y_bordered=True)
    x, w = _rnd((1, 10, 10, 96), 9).to(DEV), _rnd((96, 3, 3, 96), 10).to(DEV)
    with pytest.raises(OmniNativeError):                                   # bordered output needs bordered input
        ops.vae_conv2d(x, w, y_bordered=True)
