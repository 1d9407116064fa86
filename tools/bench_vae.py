"""Dev tool: per-op timing of one VAE decode ("VAE decode log").
   RES=1024 B=1 python tools/bench_vae.py        (RES = image side, B = images per decode call)
Wall time of decode() first, then every op wrapped in events (synchronising: the sum is a little above the free-running
time); conv lines carry their TF/s (2 * pixels * Cin * Cout * ks^2 flop)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from vllm_omni_amd import ops
from vllm_omni_amd.diffusion.models.qwen_image.autoencoder_kl_qwenimage import AutoencoderKLQwenImage
dev = torch.device("cuda:0")
vae = AutoencoderKLQwenImage(device=dev); vae.init_random_()
res = int(os.environ.get("RES", "1024")); B = int(os.environ.get("B", "1"))
z = torch.randn(B, 16, 1, res // 8, res // 8, device=dev)
# per-op timing via monkeypatching
import collections
T = collections.OrderedDict()
def wrap(name):
    f = getattr(ops, name)
    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(*a, **k); e1.record(); torch.cuda.synchronize()
        x = a[0]
        if name == "vae_conv2d":
            w = a[1]; key = f"conv {tuple(x.shape[1:3])} {w.shape[3]}->{w.shape[0]} k{w.shape[1]}" + (" up" if k.get("upsample2x") else "") + (" B" if k.get("y_bordered") else "")
            Hh, Ww = (x.shape[1] * 2, x.shape[2] * 2) if k.get("upsample2x") else x.shape[1:3]
            if k.get("x_bordered"): Hh, Ww = Hh - 2, Ww - 2
            fl = 2.0 * x.shape[0] * Hh * Ww * w.shape[3] * w.shape[0] * w.shape[1] ** 2
        else:
            key = f"{name} {tuple(x.shape) if hasattr(x, 'shape') else 'group'}"; fl = 0.0
        t, n, f0 = T.get(key, (0.0, 0, 0.0)); T[key] = (t + e0.elapsed_time(e1), n + 1, f0 + fl)
        return r
    setattr(ops, name, g)
for _ in range(2): vae.decode(z)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3): vae.decode(z)
torch.cuda.synchronize(); print(f"decode {res}^2 B={B}: {(time.time()-t0)/3*1e3:.2f} ms per call")
for n in ("vae_conv2d", "vae_rmsnorm_silu", "gemm", "softmax_rows_", "vae_upsample2x_bordered", "vae_attention"): wrap(n)
vae.decode(z)
tot = sum(v[0] for v in T.values())
for k, (t, n, fl) in T.items():
    print(f"  {k:48s} n={n:3d} {t:8.3f} ms  {fl/t/1e9 if fl else 0:8.1f} TF/s")
print("  sum of ops", f"{tot:.2f} ms")
