import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    import torch
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" else z[k]) for k in z.files}


def fp16_torso_state_dict(shapes, g):
    """Weights of tests/golden/adm_tiny_a_fp16w.npz: full-precision synthetic fp32 values with the tensors the REFERENCE's
    convert_to_fp16() cast (their names travel in the fixture) rounded to fp16 -- what a real GuidedDiffusion checkpoint holds."""
    from perceptor_amd.utils.synth import synth_state_dict
    sd = synth_state_dict(shapes, 0, rounding="none")
    for k in g["rounded_keys"].tolist():
        sd[k] = sd[k].half().float()
    chk = [float(sd[k].double().abs().sum()) for k in sorted(sd)]
    assert np.allclose(chk, g["weight_abs_sum"].numpy(), rtol=1e-6), "synthetic weights differ from the ones the fixture was generated with"
    return sd


@pytest.fixture(scope="session")
def load_golden():
    return golden
