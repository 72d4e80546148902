"""TEST INFRASTRUCTURE — runs only in the build container, never on the GPU box.

Loads the reference's hot-path source files *unmodified* from /root/reference
so golden vectors can be generated from the real code.  The
reference package's __init__ files import heavy third-party packages that are
not installed (lantern, basicsr, torchvision, open_clip, ...), so empty package
shells are registered in sys.modules first and only four tiny stand-ins are
provided for symbols the hot-path files touch at import time:

  lantern.Tensor / lantern.FunctionalBase   (type annotation + frozen record)
  basicsr.utils.download_util.load_file_from_url   (raises: no network)
  torchvision.transforms.functional          (imported, never called)
  omegaconf.listconfig.ListConfig            (an empty class: the vendored ldm UNet only asks `type(context_dim) == ListConfig`)
(gen_tokenizer additionally registers an identity `ftfy.fix_text` for ASCII prompts; no arithmetic of any path is replaced.)

Nothing from the reference is copied into this repository; only numeric
inputs/outputs are written to tests/golden/.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("PERCEPTOR_REFERENCE", "/root/reference")


def _shell(name: str, path: str | None = None) -> types.ModuleType:
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    sys.modules[name] = m
    return m


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "perceptor", "models", "guided_diffusion"))


_installed = False


def install() -> None:
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    import pydantic

    lantern = _shell("lantern")

    class Tensor(torch.Tensor):
        @classmethod
        def dims(cls, _spec):
            return torch.Tensor

    class FunctionalBase(pydantic.BaseModel):
        model_config = pydantic.ConfigDict(arbitrary_types_allowed=True, frozen=True)

        def replace(self, **kw):
            return self.model_copy(update=kw)

    lantern.Tensor = Tensor
    lantern.FunctionalBase = FunctionalBase

    _shell("basicsr", "")
    _shell("basicsr.utils", "")
    dl = _shell("basicsr.utils.download_util")

    def load_file_from_url(*a, **k):
        raise RuntimeError("offline: checkpoints are not reachable")

    dl.load_file_from_url = load_file_from_url

    tv = _shell("torchvision", "")
    tvt = _shell("torchvision.transforms", "")
    tvf = _shell("torchvision.transforms.functional")
    tv.transforms = tvt
    tvt.functional = tvf

    p = os.path.join(REF_ROOT, "perceptor")
    _shell("perceptor", p)
    _shell("perceptor.models", os.path.join(p, "models"))
    _shell("perceptor.models.guided_diffusion", os.path.join(p, "models", "guided_diffusion"))
    _shell("perceptor.models.velocity_diffusion", os.path.join(p, "models", "velocity_diffusion"))
    _shell("perceptor.models.ruclip", os.path.join(p, "models", "ruclip"))
    _shell("perceptor.models.stable_diffusion", os.path.join(p, "models", "stable_diffusion"))   # predictions.py / conditioning.py / diffusion_space.py only
    _shell("perceptor.models.slip", os.path.join(p, "models", "slip"))      # only its tokenizer.py is loaded (gen_tokenizer)
    # omegaconf (absent): the vendored ldm UNet only asks `type(context_dim) == ListConfig` (openaimodel.py:494-497): an empty class answers no
    oc = _shell("omegaconf", "")
    ocl = _shell("omegaconf.listconfig")
    ocl.ListConfig = type("ListConfig", (), {})
    oc.listconfig = ocl
    # the vendored CompVis latent-diffusion code (the original StableDiffusion UNet / VAE): oracle/gen_golden.py: gen_sd_ldm
    ld = os.path.join(p, "models", "latent_diffusion")
    for sub in ("", ".ldm", ".ldm.modules", ".ldm.modules.diffusionmodules", ".ldm.modules.distributions", ".ldm.models"):
        _shell("perceptor.models.latent_diffusion" + sub, os.path.join(ld, *sub.strip(".").split(".")) if sub else ld)
    _shell("perceptor.transforms", os.path.join(p, "transforms"))
    _shell("perceptor.transforms.resize", os.path.join(p, "transforms", "resize"))
    utils = _shell("perceptor.utils", os.path.join(p, "utils"))
    utils.cache = lambda f: f
    sys.modules["perceptor"].models = sys.modules["perceptor.models"]
    sys.modules["perceptor"].utils = utils
    _installed = True


def ref(module: str):
    """Import ``perceptor.<module>`` from the reference tree."""
    install()
    return importlib.import_module("perceptor." + module)
