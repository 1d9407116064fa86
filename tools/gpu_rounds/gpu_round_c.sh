#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_hipblaslt -o mm --output-format csv -- python $GRAFT_REPO_ROOT/tools/probe_hipblaslt.py > $GRAFT_REPO_ROOT/gpurun_out/r02c_probe.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_hipblaslt -name "*.csv" | head
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_hipblaslt/**/*kernel_trace.csv", recursive=True):
    seen = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "Cijk" in n or "gemm" in n.lower():
            key = (n, r.get("Grid_Size_X"), r.get("Workgroup_Size_X"))
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            seen.setdefault(key, []).append((d, r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size")))
    for k, v in seen.items():
        print(k[0][:400]); print("   grid", k[1], "wg", k[2], "n", len(v), "min_us", min(x[0] for x in v), "lds/vgpr/agpr/sgpr/scratch", v[0][1:])
PY
ls /opt/rocm/lib/hipblaslt/library | head -30
