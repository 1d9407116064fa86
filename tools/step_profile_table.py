#!/usr/bin/env python3
"""Per-kernel table of ONE step shape from a rocprofv3 --kernel-trace database of `tools/time_step.py <px> 60 <n> <R>`:
every launch class of a DiT block mapped to its shape (the four block GEMMs are told apart by their grid, including the tail
split's extra workgroups), average us per launch, flop per launch and the fraction of the 2.5 PF/s bf16 MFMA peak.
   python tools/step_profile_table.py <results.db> <px> <R> [cus]"""
import sqlite3
import sys

db, px, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cus = int(sys.argv[4]) if len(sys.argv) > 4 else 256
D, H, T, L = 3072, 24, 64, 60
S = (px // 16) ** 2
items = 2 * R
Mi, Mt = items * S, items * T
row_tiles = -(-Mi // 256) + -(-Mt // 256)
con = sqlite3.connect(db)
rows = con.execute("select name, grid_x, count(*), avg(duration), min(duration), sum(duration) from kernels group by name, grid_x").fetchall()
total = sum(r[5] for r in rows)


def grid_of(N, K):
    tiles = row_tiles * (N // 256)
    tt = tiles % cus
    if tiles > cus and 0 < tt and tt * 4 <= cus:           # the tail-split rule of csrc/gemm.hip tail_split_factor()
        ns = 8 if K // 64 >= 96 else 4
        return (tiles - tt + tt * ns) * 512, f"{tiles} tiles = {tiles // cus} rounds + {tt}, tail split x{ns}"
    return tiles * 512, f"{tiles} tiles = {tiles // cus} rounds + {tt}"


shapes = {"QKV (N 9216, K 3072)": (3 * D, D), "out-proj (N 3072, K 3072)": (D, D), "MLP-up + GELU (N 12288, K 3072)": (4 * D, D),
          "MLP-down (N 3072, K 12288)": (D, 4 * D)}
print(f"one denoise step = one ragged forward of {items} items x ({S} + {T}) rows at {px}^2, {L} layers; total kernel time "
      f"{total / 1e6:.1f} ms in the trace; peak 2500 TF/s")
print(f"{'launch class':44s} {'launches':>8s} {'avg us':>9s} {'min us':>9s} {'% of trace':>10s} {'TFLOP/launch':>13s} {'frac of peak':>12s}  note")
seen = set()
by_grid = {}
present = {r[1] for r in rows if "gemm_bf16_pp_kernel" in r[0]}
for name, (N, K) in shapes.items():
    g, note = grid_of(N, K)
    if g not in present:                                      # a library without the tail split (round <= 5 profiles)
        tiles = row_tiles * (N // 256)
        g, note = tiles * 512, f"{tiles} tiles = {tiles // cus} rounds + {tiles % cus}"
    by_grid.setdefault(g, []).append((name, N, K, note))
for g, lst in by_grid.items():
    rs = [r for r in rows if "gemm_bf16_pp_kernel" in r[0] and r[1] == g]
    if not rs:
        continue
    n, avg, mn, tot = sum(r[2] for r in rs), sum(r[5] for r in rs) / sum(r[2] for r in rs), min(r[4] for r in rs), sum(r[5] for r in rs)
    flop = sum(2.0 * (Mi + Mt) * N * K for _nm, N, K, _ in lst)
    label = " + ".join(x[0] for x in lst)
    per = len(lst)                                           # shapes sharing one grid: the average is over both
    print(f"{label[:44]:44s} {n:8d} {avg / 1e3:9.1f} {mn / 1e3:9.1f} {100 * tot / total:10.1f} {flop / per / 1e12:13.3f} "
          f"{flop / per / (avg * 1e-9) / 2.5e15:12.3f}  {lst[0][3]}" + (" (two shapes, one grid: pair average)" if per > 1 else ""))
    seen.update(id(r) for r in rs)
att = [r for r in rows if "flash_attn_fwd" in r[0]]
if att:
    n, tot = sum(r[2] for r in att), sum(r[5] for r in att)
    comb = [r for r in rows if "attn_split_combine" in r[0]]
    ctot = sum(r[5] for r in comb)
    flop = items * H * 4.0 * (S + T) ** 2 * 128
    avg = (tot + ctot) / n
    print(f"{'joint attention (+ split combine)':44s} {n:8d} {avg / 1e3:9.1f} {min(r[4] for r in att) / 1e3:9.1f} {100 * (tot + ctot) / total:10.1f} "
          f"{flop / 1e12:13.3f} {flop / (avg * 1e-9) / 2.5e15:12.3f}  grid {att[0][1] // 256} workgroups" + (f", combine {ctot / max(1, sum(r[2] for r in comb)) / 1e3:.1f} us" if comb else ""))
for key, label in (("rownorm_kernel", "AdaLN / RMSNorm (rownorm_kernel)"), ("gemm_tail_finish", "GEMM tail-split finish"),
                   ("gemm_splitk_finish", "GEMM split-K finish"), ("splitk_finish_adaln", "split-K finish + AdaLN (fused, ABI v13)"),
                   ("linear_smallbatch", "modulation / timestep GEMVs")):
    rs = [r for r in rows if key in r[0]]
    if rs:
        n, tot = sum(r[2] for r in rs), sum(r[5] for r in rs)
        print(f"{label:44s} {n:8d} {tot / n / 1e3:9.1f} {min(r[4] for r in rs) / 1e3:9.1f} {100 * tot / total:10.1f}")
other = [r for r in rows if not any(k in r[0] for k in ("gemm_bf16_pp_kernel", "flash_attn_fwd", "attn_split_combine", "rownorm_kernel",
                                                          "gemm_tail_finish", "gemm_splitk_finish", "splitk_finish_adaln", "linear_smallbatch"))]
small = [r for r in rows if "gemm_bf16_pp_kernel" in r[0] and id(r) not in seen]
print(f"{'other GEMM launches (img_in, txt_in, proj_out)':44s} {sum(r[2] for r in small):8d} {'':9s} {'':9s} {100 * sum(r[5] for r in small) / total:10.1f}")
print(f"{'everything else (torch init / RNG, gathers)':44s} {sum(r[2] for r in other):8d} {'':9s} {'':9s} {100 * sum(r[5] for r in other) / total:10.1f}")
