from .diffusers_loader import DiffusersPipelineLoader  # noqa: F401
