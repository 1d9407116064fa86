#!/bin/bash
# Round 5, first GPU call (run on the box through gpurun): (1) the queued GEMM experiments of tools/probe/next_gemm_experiments.sh,
# (2) the fp8 instance on the restructured loop through the fp8 GPU tests, (3) rocprofv3 kernel-trace of the bench command and the
# three PMC passes of the roofline launch on the SHIPPED library (round 4's profiles describe its predecessor).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
L=vllm_omni_amd/csrc/build/abl
P=vllm_omni_amd/libomni_cdna4.so
(
  timeout 60 ./tools/probe/pp_probe --sweep $P $L/libomni_sched0.so $L/libomni_big.so
  timeout 40 ./tools/probe/pp_probe --iters 20 $L/libomni_sched0.so $P $L/libomni_big.so $L/libomni_p_base.so $L/libomni_p_big.so $P $L/libomni_big.so
  for s in "--n 3072 --k 3072 --epi 2" "--n 9216 --k 3072 --epi 4" "--n 3072 --k 12288 --epi 2"; do
    echo "== $s"
    timeout 30 ./tools/probe/pp_probe --iters 20 $s $L/libomni_sched0.so $P $L/libomni_big.so $P $L/libomni_big.so
  done
  timeout 20 ./tools/probe/attn_bench --iters 20 $P
) > $OUT/r05_gemm_first.log 2>&1
# fp8 on the new loop: the fp8 GPU tests against the variant library
OMNI_DEV_LIB=$L/libomni_fp8new.so timeout 300 python -c "import tools.devlib, pytest, sys; sys.exit(pytest.main(['tests/test_gpu_fp8.py', '-m', 'gpu', '-x', '-q']))" > $OUT/r05_fp8_sched17_pytest.log 2>&1
# profile of the shipped library
timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_r05a_bench -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-engine > $OUT/prof_r05a_bench.log 2>&1
for K in roofline; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_r05a_${K}_sq -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_r05a_${K}_sq.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_r05a_${K}_fetch -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_r05a_${K}_fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_r05a_${K}_write -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_r05a_${K}_write.log 2>&1
done
tail -60 $OUT/r05_gemm_first.log
tail -5 $OUT/r05_fp8_sched17_pytest.log
tail -3 $OUT/prof_r05a_bench.log
find $OUT/prof_r05a_bench $OUT/pmc_r05a_* -name "*.csv" | head
