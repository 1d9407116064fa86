export TMPDIR=/tmp
L=gpurun_out/r06_ab_scc_clobber.log; : > $L
for rep in 1 2 3; do
  for lib in product noscc; do
    if [ $lib = product ]; then unset OMNI_DEV_LIB; else export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_noscc.so; fi
    echo "$lib rep $rep: $(python tools/time_step.py 1024 60 3 5 2>&1 | tail -1 | cut -c1-150)" >> $L
  done
done
unset OMNI_DEV_LIB
python tools/time_config1.py 2>&1 | tail -1 >> $L
rocprofv3 --kernel-trace --stats -T -d gpurun_out/prof_r06b_config1 -o p -- python tools/run_config1.py 5 > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out r06b 2>&1 | head -12 >> $L
cat $L
