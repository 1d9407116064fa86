#!/usr/bin/env python3
"""dev tool: time ONE kernel of several library builds (tools/build_variants.sh) in interleaved subprocess rounds.
   python tools/bench_libs.py attention|gemm|qkv lib1.so lib2.so ...   (each lib is run AB_ROUNDS times, round-robin)
   a lib may carry environment knobs: path/lib.so@OMNI_ATTN_MFMA=32@OTHER=1"""
import os
import statistics
import subprocess
import sys

which, libs = sys.argv[1], sys.argv[2:]
rounds = int(os.environ.get("AB_ROUNDS", "3"))
code = {
    "attention": r'''
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from vllm_omni_amd import ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
H, S, B = 24, 4160, int(os.environ.get("AB_ATTN_B", "6"))
q, k, v = ((torch.randn(B * S, H * 128, device=dev, generator=g)).to(torch.bfloat16) for _ in range(3))
cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(dev)
o = ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128))
q4, k4, v4 = (x.view(B, S, H, 128).permute(0, 2, 1, 3).float() for x in (q, k, v))
ref = torch.nn.functional.scaled_dot_product_attention(q4[5:, 20:], k4[5:, 20:], v4[5:, 20:])
err = float((o.view(B, S, H, 128).permute(0, 2, 1, 3)[5:, 20:].float() - ref).norm() / ref.norm())
t = timeit(lambda: ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128)), iters=10)
print("RESULT", 4.0 * B * H * S * S * 128 / t / 1e12, err)
''',
    "gemm": r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from vllm_omni_amd import ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
D = 3072; Mi, Mt, N, K = 24576, 384, 4 * D, D
rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
xi, xt = ops.w_to_k32_blocked(rn(Mi, K)), ops.w_to_k32_blocked(rn(Mt, K))
wi, wt, b = ops.w_to_k32_blocked(rn(N, K, sc=0.02)), ops.w_to_k32_blocked(rn(N, K, sc=0.02)), rn(N)
oi, ot = torch.empty(Mi, N, dtype=torch.bfloat16, device=dev), torch.empty(Mt, N, dtype=torch.bfloat16, device=dev)
fn = lambda: ops.gemm([ops.GemmGroupArgs(xi, wi, b, oi, a_k32_blocked=True, out_k32_blocked=True),
                       ops.GemmGroupArgs(xt, wt, b, ot, a_k32_blocked=True, out_k32_blocked=True)], ops.EPI_BIAS_GELU_TANH, w_k32_blocked=True)
t = timeit(fn, iters=10)
print("RESULT", 2.0 * (Mi + Mt) * N * K / t / 1e12, 0.0)
''',
    "qkv": r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from vllm_omni_amd import ops
from vllm_omni_amd.diffusion.batch import build_ragged_batch
from vllm_omni_amd.diffusion.models.qwen_image.rope import rope_table
from tools.bench_kernels import timeit
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
BF16 = torch.bfloat16
D, H, T, grid, items = 3072, 24, 64, (1, 64, 64), 10
rb = build_ragged_batch([T] * items, grid, list(range(items)), T)
maps = rb.device_maps(dev)
Ri, Rt, Rj = rb.n_img_rows, rb.n_txt_rows, rb.n_joint_rows
cos, sin = rope_table(grid, T)
cosb, sinb = cos.to(dev, BF16), sin.to(dev, BF16)
rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(BF16)
xi, xt = ops.w_to_k32_blocked(rn(Ri, D)), ops.w_to_k32_blocked(rn(Rt, D))
wi, wt = ops.w_to_k32_blocked(rn(3 * D, D, sc=0.02)), ops.w_to_k32_blocked(rn(3 * D, D, sc=0.02))
bi, bt = rn(3 * D, sc=0.1), rn(3 * D, sc=0.1)
nw = [(1 + rn(128, sc=0.1).float()).to(BF16) for i in range(4)]
jp = maps["joint_pos"]
q = torch.zeros(Rj, D, dtype=BF16, device=dev); k, v = torch.zeros_like(q), torch.zeros_like(q)
kw_i = dict(qk_norm_q_w=nw[0], qk_norm_k_w=nw[1], qk_rope_cos=cosb, qk_rope_sin=sinb, qk_row_pos=jp[maps["img_joint_row"].long()].contiguous())
kw_t = dict(qk_norm_q_w=nw[2], qk_norm_k_w=nw[3], qk_rope_cos=cosb, qk_rope_sin=sinb, qk_row_pos=jp[maps["txt_joint_row"].long()].contiguous())
fn = lambda: ops.gemm([ops.GemmGroupArgs(xi, wi, bi, q, out1=k, out2=v, out_row_map=maps["img_joint_row"], a_k32_blocked=True, **kw_i),
                       ops.GemmGroupArgs(xt, wt, bt, q, out1=k, out2=v, out_row_map=maps["txt_joint_row"], a_k32_blocked=True, **kw_t)],
                      ops.EPI_BIAS_SPLIT3_QKNORM_ROPE, split_n=D, w_k32_blocked=True)
t = timeit(fn, iters=10)
print("RESULT", 2.0 * (Ri + Rt) * 3 * D * D / t / 1e12, float(q.float().abs().mean()))
''',
}[which]
res = {l: [] for l in libs}
for _ in range(rounds):
    for l in libs:
        path, *knobs = l.split("@")
        env = dict(os.environ, OMNI_DEV_LIB=os.path.abspath(path), **dict(kv.split("=", 1) for kv in knobs))
        out = subprocess.run([sys.executable, "-c", "import tools.devlib\n" + code], env=env, capture_output=True, text=True)
        line = [x for x in out.stdout.splitlines() if x.startswith("RESULT")]
        if not line:
            print(l, "FAILED", out.stderr[-400:])
            continue
        _, tf, err = line[0].split()
        res[l].append((float(tf), float(err)))
for l, v in res.items():
    if v:
        print(f"{which} {os.path.basename(l):28s} median {statistics.median(x[0] for x in v):7.1f} TF/s  all {[round(x[0]) for x in v]}  err {v[0][1]:.2e}")
