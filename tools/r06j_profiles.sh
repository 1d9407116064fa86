#!/bin/bash
# Profiles of the library that ships (rocprofv3 kernel trace of the bench command, PMC passes of the roofline / attention launches,
# per-kernel traces of the step shapes) — the test suite and the default bench run of the same library: tools/r06e_verify.sh.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r06}
: # (tests + default bench: tools/r06e_verify.sh on the same library)

timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_${TAG}_bench -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-engine > $OUT/prof_${TAG}_bench.log 2>&1
for K in roofline attention10; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_${TAG}_${K}_sq -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_sq.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_${TAG}_${K}_fetch -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_${TAG}_${K}_write -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_write.log 2>&1
done
for spec in "2048 1" "1024 1" "1024 2"; do
  set -- $spec
  timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_${TAG}_final_step_$1px_R$2 -o p -- python tools/time_step.py $1 60 2 $2 > $OUT/prof_${TAG}_final_step_$1px_R$2.log 2>&1
  python tools/step_profile_table.py $OUT/prof_${TAG}_final_step_$1px_R$2/*.db $1 $2 > $OUT/${TAG}_step_table_$1px_R$2.txt 2>&1
  tail -1 $OUT/prof_${TAG}_final_step_$1px_R$2.log | grep -v amdgpu >> $OUT/${TAG}_step_table_$1px_R$2.txt
done
timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_${TAG}_final_config1 -o p -- python tools/run_config1.py 5 > /dev/null 2>&1
python tools/time_config1.py cold 2>&1 | tail -1 > $OUT/${TAG}_config1_ms.txt
python tools/time_config1.py warm 2>&1 | tail -1 >> $OUT/${TAG}_config1_ms.txt
timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_${TAG}_final_step_256px_R1 -o p -- python tools/time_step.py 256 60 5 1 > $OUT/prof_${TAG}_final_step_256px_R1.log 2>&1
python tools/step_profile_table.py $OUT/prof_${TAG}_final_step_256px_R1/*.db 256 1 > $OUT/${TAG}_step_table_256px_R1.txt 2>&1
OMNI_PROFILES_DIR=$OUT python tools/summarize_prof.py $OUT $TAG > $OUT/${TAG}_rocprof_summary_final.txt 2>&1
# gpurun merges at most 64 MiB back: the counter and step databases are summarised above; only the bench trace travels
rm -rf $OUT/pmc_${TAG}_* $OUT/prof_${TAG}_final_* 
find $OUT -name "*.db" -size +45M -delete
sha256sum vllm_omni_amd/libomni_cdna4.so | cut -c1-16 > $OUT/${TAG}_library_sha.txt
cat $OUT/${TAG}_step_table_2048px_R1.txt $OUT/${TAG}_config1_ms.txt
