"""perceptor_amd — MI355X-native guided-diffusion sampling hot path behind perceptor's Python surface.

    from perceptor_amd import models, losses     # instead of: from perceptor import models, losses
"""
from . import models  # noqa: F401  (first: losses imports models)
from . import losses, transforms  # noqa: F401

__version__ = "0.1.0"
