from .stable_diffusion import StableDiffusion  # noqa: F401
from .predictions import Predictions  # noqa: F401
from .conditioning import Conditioning  # noqa: F401
