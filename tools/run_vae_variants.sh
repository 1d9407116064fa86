#!/bin/bash
# Dev tool (GPU box): VAE op tests, then tools/bench_vae.py for the product library and for every dev variant named on the
# command line (vllm_omni_amd/csrc/build/abl/libomni_<name>.so from tools/build_variants.sh).  Logs: gpurun_out/<tag>/.
tag=${TAG:-vae_variants}
mkdir -p gpurun_out/$tag
timeout 600 python -m pytest tests/test_gpu_vae_ops.py tests/test_gpu_pipeline.py tests/test_gpu_bench_shape_parity.py tests/test_gpu_edit.py -k "vae or decode or conv or upsample or softmax" -x -q > gpurun_out/$tag/pytest.log 2>&1; tail -3 gpurun_out/$tag/pytest.log
for res in ${RESES:-1024}; do
  RES=$res timeout 100 python tools/bench_vae.py 2>/dev/null > gpurun_out/$tag/vae_product_$res.log; head -1 gpurun_out/$tag/vae_product_$res.log
done
for v in "$@"; do
  OMNI_DEV_LIB=vllm_omni_amd/csrc/build/abl/libomni_$v.so RES=1024 timeout 100 python -c "import tools.devlib, runpy; runpy.run_path('tools/bench_vae.py', run_name='__main__')" 2>/dev/null > gpurun_out/$tag/vae_$v.log
  echo $v; head -1 gpurun_out/$tag/vae_$v.log
done
