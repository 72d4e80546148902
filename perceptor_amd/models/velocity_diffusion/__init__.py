from .predictions import Predictions
from .velocity_diffusion import VelocityDiffusion
