"""Picklable pipeline factory for the GPU engine tests: a SMALL random-weight Qwen-Image pipeline (2 DiT layers, 2 heads,
joint dim 128) with a random-weight Qwen2.5-VL text model of the same width and the full-size VAE decoder."""
import torch


def make_small_pipeline():
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.models.qwen_image.autoencoder_kl_qwenimage import AutoencoderKLQwenImage
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel
    from vllm_omni_amd.diffusion.models.qwen_image.text_encoder import QwenPromptEncoder

    dev = torch.device("cuda", torch.cuda.current_device())
    tr = QwenImageTransformer2DModel(num_layers=2, num_attention_heads=2, joint_attention_dim=128, device=dev)
    tr.init_random_(seed=11)
    vae = AutoencoderKLQwenImage(device=dev).init_random_(seed=12)
    enc = QwenPromptEncoder.random_init(hidden_size=128, num_layers=2, num_heads=4, num_kv_heads=2, intermediate_size=256,
                                        device=dev, dtype=torch.bfloat16, seed=13)
    return QwenImagePipeline(od_config=OmniDiffusionConfig(max_step_batch=4), device=dev, transformer=tr, vae=vae,
                             text_encoder=enc)
