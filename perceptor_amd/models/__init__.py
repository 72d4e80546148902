"""Drop-in for ``perceptor.models`` on the guided-diffusion hot path."""
from .guided_diffusion import GuidedDiffusion
from .open_clip import CLIP, OpenCLIP
from .velocity_diffusion import VelocityDiffusion
from .stable_diffusion import StableDiffusion
from .transformers_openai_clip import TransformersOpenAICLIP
