export TMPDIR=/tmp
L=$PWD/gpurun_out/r06_ab_vs_round5_library.log; : > $L
for rep in 1 2 3; do
  for tree in r06 r05; do
    if [ $tree = r06 ]; then d=$PWD; else d=$PWD/.gpu_scratch; fi
    for spec in "1024 5" "2048 1" "1024 1"; do
      set -- $spec
      echo "$tree rep $rep: $(cd $d && python tools/time_step.py $1 60 3 $2 2>&1 | tail -1 | cut -c1-170)" >> $L
    done
  done
done
cat $L
