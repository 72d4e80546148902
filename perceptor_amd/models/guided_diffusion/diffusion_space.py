"""images in [0,1] <-> x in [-1,1] (reference: guided_diffusion/diffusion_space.py:1-6)."""


def encode(images):
    return images.mul(2).sub(1)


def decode(x):
    return x.add(1).div(2)
