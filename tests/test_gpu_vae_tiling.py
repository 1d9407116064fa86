"""GPU: the VAE's memory savers — `use_slicing` (one latent / image per pass) and `use_tiling` (overlapping spatial tiles,
cross-faded) — which od_config.vae_use_slicing / vae_use_tiling switch on (reference data.py:299-300, registry.py:88-92;
autoencoder_kl_qwenimage.py:742-773 switches, :905-969 tiled_encode, :971-1031 tiled_decode, :889-903 blends).

Round 4 refused `vae_use_tiling` (a 4096^2 request had no path).  Checker: the oracle's tiled decode / encode, pinned to a RUN of
the reference's vendored VAE with `enable_tiling()` (tests/golden/vae_tiled_36x40_fp32.npz, tests/test_oracle_golden.py); the
stitching arithmetic itself is bit-equal to the reference's loops (tests/test_host_logic.py).  Tolerances: the decoder's image bar
(rel_l2 <= 3e-2, mean |err| <= 2e-2: tests/e2e/offline_inference/test_sequence_parallel.py:128-147), encoder mean rel_l2 <= 2e-2."""
import time

import pytest
import torch

import qwen_image_oracle as O
from _util import load_golden, rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"


def _vae(with_encoder=True):
    from vllm_omni_amd.diffusion.models.qwen_image.autoencoder_kl_qwenimage import AutoencoderKLQwenImage

    vae = AutoencoderKLQwenImage(device=DEV, with_encoder=with_encoder)
    Pd, Pe = O.make_vae_params(), O.make_vae_encoder_params()
    vae.load_weights(list(Pd.items()) + (list(Pe.items()) if with_encoder else []))
    return vae, Pd, Pe


def test_tiled_decode_and_encode_match_the_reference_run():
    z, meta, c = load_golden("vae_tiled_36x40_fp32")
    vae, _, _ = _vae()
    vae.enable_tiling()
    img = vae.decode(torch.from_numpy(z["z"]).to(DEV, BF16))[0]
    img_in = torch.rand(1, 3, 1, 288, 320, generator=torch.Generator().manual_seed(c["seeds"][1])) * 2 - 1
    mean = vae.encode(img_in.to(DEV, BF16))
    torch.cuda.synchronize()
    ref, refm = torch.from_numpy(z["image"]), torch.from_numpy(z["mean"])
    r, d = rel_l2(img, ref), float((img.float().cpu() - ref).abs().mean())
    rm = rel_l2(mean, refm)
    print(f"tiled decode 288x320 vs the reference run: rel_l2 {r:.3e} mean|err| {d:.3e}; |max| {float(img.float().abs().max()):.2f} "
          f"(un-clamped, reference {float(ref.abs().max()):.2f}); tiled encode mean rel_l2 {rm:.3e}")
    assert img.shape == ref.shape and r <= 3e-2 and d <= 2e-2
    assert float(img.float().abs().max()) > 1.5                       # the tiled path does not clamp (reference :844-845 vs :857)
    assert mean.shape == refm.shape and rm <= 2e-2
    # below the tile size nothing changes: the plain (clamped) path
    small = torch.randn(1, 16, 1, 16, 24, generator=torch.Generator().manual_seed(2)).to(DEV, BF16)
    a = vae.decode(small)[0]
    vae.disable_tiling()
    assert torch.equal(a, vae.decode(small)[0]) and float(a.float().abs().max()) <= 1.0


def test_slicing_equals_one_pass_per_item():
    vae, _, _ = _vae()
    z = (torch.randn(3, 16, 1, 24, 16, generator=torch.Generator().manual_seed(3)) * 1.5).to(DEV, BF16)
    img = (torch.rand(3, 3, 1, 64, 96, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(DEV, BF16)
    one_by_one = torch.cat([vae.decode(z[i:i + 1])[0] for i in range(3)])
    enc_one = torch.cat([vae.encode(img[i:i + 1]) for i in range(3)])
    vae.enable_slicing()
    sliced, enc_sliced = vae.decode(z)[0], vae.encode(img)
    torch.cuda.synchronize()
    assert torch.equal(sliced, one_by_one) and torch.equal(enc_sliced, enc_one)


def test_tiled_decode_at_1536px_matches_the_oracle_and_4096px_has_a_path():
    """8 x 8 tiles (192 x 192 latent) against the fp32 oracle's tiled decode on the GPU; then the request round 4 had no path
    for: a 512 x 512 latent -> 4096 x 4096 pixels, 22 x 22 tiles."""
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    vae, Pd, _ = _vae(with_encoder=False)
    vae.enable_tiling()
    Pg = {k: v.to(BF16).float().to(DEV) for k, v in Pd.items()}
    z = (torch.randn(1, 16, 1, 192, 192, generator=torch.Generator().manual_seed(6)) * 1.5).to(BF16)
    img = vae.decode(z.to(DEV))[0]
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.vae_tiled_decode(Pg, z.float().to(DEV))
    r, d = rel_l2(img, ref), float((img.float() - ref).abs().mean())
    print(f"tiled decode 1536px (64 tiles) vs fp32 oracle: rel_l2 {r:.3e} mean|err| {d:.3e}")
    assert img.shape == ref.shape == (1, 3, 1, 1536, 1536) and r <= 3e-2 and d <= 2e-2
    del ref
    big = (torch.randn(1, 16, 1, 512, 512, generator=torch.Generator().manual_seed(7)) * 1.5).to(DEV, BF16)
    vae.decode(big[:, :, :, :64, :64])                                # warm-up of the tile shapes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = vae.decode(big)[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"tiled decode 4096 x 4096 (484 tiles): {dt * 1e3:.0f} ms")
    assert out.shape == (1, 3, 1, 4096, 4096) and torch.isfinite(out.float()).all() and float(out.float().std()) > 0.05
