"""Drop-in for ``perceptor.losses`` on the guided-diffusion hot path."""
from .interface import LossInterface
from .open_clip import CLIP, OpenCLIP
