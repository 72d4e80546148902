from .guided_diffusion import GuidedDiffusion
from .predictions import Predictions
