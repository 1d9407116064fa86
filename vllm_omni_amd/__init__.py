"""vllm_omni_amd — MI355X-native (gfx950) drop-in for the Qwen-Image DiT denoising path of vllm-omni.

Package layout mirrors the reference's `vllm_omni/diffusion/` so that the same import paths / class
names resolve:  layers/ (CustomOp plug-in ops), attention/ (backend registry),
models/qwen_image/ (transformer, pipeline, VAE), worker/ (one process per GPU), distributed/ (DP group).
All arithmetic of the hot path runs in libomni_cdna4.so (csrc/, C-ABI in include/omni_cdna4.h).
"""
__version__ = "0.1.0"
