"""Dev tool: attention throughput against the sequence length / batch (attention (iv)): "useful" counts
the real rows, "issued" counts the 256-row blocks the 64-queries-per-wave kernel launches (a 4160-row item issues 17 blocks
for 16.25 blocks of rows)."""
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from vllm_omni_amd import ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
H = 24
for S, B in ((4096, 10), (4160, 10), (4352, 10), (4096, 32), (4160, 32), (4096, 16), (4160, 16), (4160, 10), (4096, 10)):
    q, k, v = ((torch.randn(B * S, H * 128, device=dev, generator=g)).to(torch.bfloat16) for _ in range(3))
    cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(dev)
    t = timeit(lambda: ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128)), iters=8)
    nb = B * H * ((S + 255) // 256)
    print(f"S={S} B={B}: {t*1e3:7.3f} ms  blocks {nb} = {nb/256:.2f} rounds  {t*1e6/(nb/256):7.2f} us/round  {4.0*B*H*S*S*128/t/1e12:7.1f} TF/s (useful)  {4.0*nb*256*((S+63)//64*64)*128/t/1e12:7.1f} TF/s (issued)", flush=True)
    del q, k, v
