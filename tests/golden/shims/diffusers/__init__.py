"""minimal stand-in: see ../README.md"""
from .pipelines.stable_diffusion.pipeline_stable_diffusion import StableDiffusionPipeline  # noqa: F401
