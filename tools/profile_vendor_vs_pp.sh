#!/bin/bash
# Dev tool (GPU box, through gpurun): the SAME counter passes over the product's roofline launch and over the vendor kernel
# (torch.mm -> hipBLASLt) on that shape, in one call = one box.  Separate --pmc passes with --kernel-trace only.
# usage: tools/profile_vendor_vs_pp.sh <tag>;  summary: python tools/summarize_pmc_pairs.py gpurun_out <tag>
TAG=${1:-r04v}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
for K in roofline torchmm_mlp_up; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc_${TAG}_${K}_sq -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_sq.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_${TAG}_${K}_fetch -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_${TAG}_${K}_write -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_write.log 2>&1
done
TAG=$TAG python - <<'PY' > $OUT/pmc_${TAG}_summary.txt 2>&1
import glob, sqlite3, os
root = os.environ.get("OUT", "gpurun_out")
for f in sorted(glob.glob(f"gpurun_out/pmc_%s_*/*.db" % os.environ.get("TAG", "r04v"))):
    con = sqlite3.connect(f)
    print("==", f)
    try:
        rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                           "group by kernel_name, counter_name order by avg(duration) desc").fetchall()
    except Exception as e:
        print("  ", e); continue
    for k, c, n, v, d in rows:
        if d and d > 500e3:
            print(f"   {k[:90]:90s} {c:26s} n={n:3d} mean {v:16.1f} avg_dur_us {d/1e3:9.2f}")
PY
