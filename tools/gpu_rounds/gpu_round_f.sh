#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/profile_round.sh r02 > gpurun_out/r02f_profile.log 2>&1; echo "profile rc=$?"
python tools/summarize_prof.py gpurun_out r02 > gpurun_out/r02_rocprof_summary.txt 2>&1; head -45 gpurun_out/r02_rocprof_summary.txt
cp profiles/r02_roofline_traffic_pmc.json gpurun_out/ 2>/dev/null
grep -E "gemm_bf16_pp|flash_attn" gpurun_out/r02_rocprof_summary.txt | grep -E "TCC|FETCH|WRITE|MFMA|WAVE_CYCLES|WAIT" | head -40
