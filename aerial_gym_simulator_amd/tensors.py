"""SoA storage helpers.

Device state is stored component-major ([C, N], what the kernels read with perfectly
coalesced loads); callers see the reference's [N, C] shapes through transposed views that
alias the same memory, so in-place writes from user code land in the SoA buffers.
"""
import torch


def soa(channels, num_envs, device, dtype=torch.float32, fill=0.0):
    t = torch.full((channels, num_envs), fill, dtype=dtype, device=device)
    return t


def aos_view(soa_tensor, start=0, stop=None):
    """[N, stop-start] view of rows start:stop of a [C, N] tensor (shares memory)."""
    return soa_tensor[start:stop].t()
