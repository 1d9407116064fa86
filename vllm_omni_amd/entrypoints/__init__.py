from .omni_diffusion import OmniDiffusion

__all__ = ["OmniDiffusion"]
