#!/usr/bin/env python3
"""Condense rocprofv3 rocpd (SQLite) outputs under gpurun_out/ into small text summaries for profiles/.
   python tools/summarize_prof.py gpurun_out r01"""
import glob
import sqlite3
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"


def short(n):
    return n.replace("(anonymous namespace)::", "")[:100]


for f in sorted(glob.glob(f"{root}/prof_{tag}_*/*.db")):
    con = sqlite3.connect(f)
    print(f"== kernel stats (rocprofv3 --kernel-trace --stats): {f}")
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"   total kernel time {tot/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
    for name, n, s, a, mn, mx in rows[:30]:
        print(f"   {short(name):100s} calls {n:6d} total_ms {s/1e6:10.2f} avg_us {a/1e3:9.2f} min_us {mn/1e3:9.2f} max_us {mx/1e3:9.2f} pct {100*s/tot:5.1f}")
    # GEMM launches by grid size (distinguishes shapes sharing one template instance)
    rows = con.execute("select name, grid_x, count(*), avg(duration) from kernels where name like '%gemm_bf16%' "
                       "group by name, grid_x order by name, grid_x").fetchall()
    for name, gx, n, a in rows:
        print(f"   [gemm by grid] {short(name)[:60]:60s} grid_x {gx:8d} calls {n:6d} avg_us {a/1e3:9.2f}")
for f in sorted(glob.glob(f"{root}/pmc_{tag}_*/*.db")):
    con = sqlite3.connect(f)
    print(f"== counters: {f}")
    try:
        rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                           "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    except Exception as e:  # noqa: BLE001
        print("   ", e)
        continue
    for k, c, n, v, d in rows:
        if "gemm" in k or "attn" in k:
            print(f"   {short(k)[:70]:70s} {c:28s} n={n:3d} mean {v:18.1f}  avg_dur_us {d/1e3 if d else 0:9.2f}")

# roofline-kernel fabric traffic per launch -> JSON that bench.py reports as roofline.traffic
# (MI355X_MICROARCH.md "HBM": FETCH_SIZE is in KB and counts 128-B requests at 64 B on gfx950 -> x2; WRITE_SIZE in KB, uncalibrated)
import json
vals = {}
for f in sorted(glob.glob(f"{root}/pmc_{tag}_roofline_*/*.db")):
    con = sqlite3.connect(f)
    try:
        for k, c, v in con.execute("select kernel_name, counter_name, avg(value) from counters_collection "
                                   "where kernel_name like '%gemm_bf16_pp%' or kernel_name like '%gemm_bf16_ring%' group by kernel_name, counter_name"):
            vals[c] = v
    except Exception:  # noqa: BLE001
        pass
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    import os
    R = int(os.environ.get("BENCH_R", "5"))
    out = {"kernel": f"gemm_bf16_pp_kernel<GELU> M={2 * R * 4096}+{2 * R * 64} N=12288 K=3072 (bench.py roofline launch, R={R})",
           "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
           "traffic_bytes": 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024,
           "note": "fabric-side bytes per launch = 2 x FETCH_SIZE (gfx950 half-count correction) + WRITE_SIZE; "
                   "Infinity-Cache hits are included, so this bounds HBM traffic from above",
           "tcc_hit": vals.get("TCC_HIT_sum"), "tcc_miss": vals.get("TCC_MISS_sum")}
    dst = os.path.join(os.environ.get("OMNI_PROFILES_DIR", "profiles"), f"{tag}_roofline_traffic_pmc.json")
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", dst, out["traffic_bytes"] / 1e6, "MB")
