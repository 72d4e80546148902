"""Drop-in for ``perceptor.losses`` on the guided-diffusion hot path."""
from .open_clip import CLIP, LossInterface, OpenCLIP
from .velocity_diffusion import VelocityDiffusion
