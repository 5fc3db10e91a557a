"""Where every random number of the simulator comes from.

The reference draws with torch's global generator on the sim device and with python's
`random` for the sub-step count (env_manager.py:417-425).  Keeping the draws behind one
small object lets tests replay recorded streams bit-for-bit (tests/ only) while the
kernels stay deterministic functions of their inputs.
"""
import random

import torch


class TorchRandomSource:
    def __init__(self, device, generator=None):
        self.device = torch.device(device)
        self.generator = generator

    def rand(self, *shape, tag=""):
        return torch.rand(*shape, device=self.device, generator=self.generator)

    def rand_into(self, out, tag=""):
        if self.generator is None:
            return out.uniform_(0.0, 1.0)
        return out.uniform_(0.0, 1.0, generator=self.generator)

    def bernoulli(self, p, *shape, tag=""):
        return torch.bernoulli(torch.full(shape, float(p), device=self.device), generator=self.generator)

    def normal_into(self, out, tag=""):
        return out.normal_(0.0, 1.0, generator=self.generator)

    def gauss(self, mean, std):
        return random.gauss(mean, std)
