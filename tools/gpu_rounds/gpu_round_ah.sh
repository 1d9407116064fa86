#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r02ah_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r02ah_pytest.log
timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_r02ah_c1 -o c1 -- python tools/run_config1.py 3 > $OUT/prof_r02ah_c1.log 2>&1; echo rc=$?
python tools/summarize_prof.py gpurun_out r02ah 2>&1 | grep -v "^W2026" | head -30
for i in 1 2; do python - <<'PY'
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import subprocess
PY
done
OMNI_GEMM_SPLITK=0 python tools/run_config1.py 1 > /dev/null 2>&1
python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
for knob in ("0", "1"):
    import subprocess
    code = '''
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
from vllm_omni_amd.diffusion.request import OmniDiffusionRequest
dev = torch.device("cuda:0")
pipe = QwenImagePipeline(od_config=OmniDiffusionConfig(model="x"), device=dev)
pipe.transformer.init_random_(seed=1234); pipe.vae.init_random_(seed=4321)
g = torch.Generator().manual_seed(3)
req = OmniDiffusionRequest(height=256, width=256, num_inference_steps=4, true_cfg_scale=4.0,
    latents=torch.randn(1, 256, 64, generator=g).to(dev, torch.bfloat16), prompt_embeds=torch.randn(1, 64, 3584, generator=g).to(dev, torch.bfloat16),
    negative_prompt_embeds=torch.randn(1, 64, 3584, generator=g).to(dev, torch.bfloat16), output_type="latent")
f = lambda: pipe.decode_latents(pipe.generate([req], output_type="latent")[0].output, 256, 256)
f(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): f()
torch.cuda.synchronize(); print("config1 ms/image", (time.perf_counter() - t0) / 5 * 1e3)
'''
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMNI_GEMM_SPLITK=knob), capture_output=True, text=True)
    print("OMNI_GEMM_SPLITK=" + knob, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
PY
