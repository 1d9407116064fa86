#!/bin/bash
# Dev tool (GPU box): SQ / fabric counters of the VAE's bordered conv at the decoder's shapes (separate --pmc passes).
# usage: tools/profile_conv.sh <tag> [shape ...]      shape = <Cin>_<Cout>_<side>[_n]
TAG=${1:-conv}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
for K in ${@:-384_384_128 384_384_256 192_192_512 96_96_1024}; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_${TAG}_conv${K}_sq -o pmc -- python tools/run_kernel.py conv$K 5 > $OUT/pmc_${TAG}_conv${K}_sq.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES -d $OUT/pmc_${TAG}_conv${K}_f -o pmc -- python tools/run_kernel.py conv$K 5 > $OUT/pmc_${TAG}_conv${K}_f.log 2>&1
done
python3 - <<'PY'
import glob, sqlite3
for f in sorted(glob.glob("gpurun_out/pmc_*conv*/*.db")):
    con = sqlite3.connect(f)
    print("==", f)
    try:
        for k, c, n, v, d in con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%conv_bordered%' group by kernel_name, counter_name"):
            print(f"   {c:28s} n={n} mean {v:16.1f} dur_us {d/1e3:8.1f}")
    except Exception as e:
        print("  ", e)
PY
