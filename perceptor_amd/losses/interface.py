import torch


class LossInterface(torch.nn.Module):
    """perceptor/losses/interface.py:4-6"""

    def forward(self, images):
        raise NotImplementedError
