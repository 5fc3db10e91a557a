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


class TensorDict(dict):
    """The global tensor dict (the reference's plain dict, env_manager.py:83) with ONE addition: a key can carry a hook that
    runs before its value is handed out.  The lean step (`args={"lean_step": True}`, AGX_LAUNCH_LEAN) uses it to recompute
    the derived tensors it no longer stores every step the moment somebody looks at them, and to refuse the ones it cannot
    reconstruct.  Without hooks it behaves like the dict it is."""

    __slots__ = ("_hooks",)

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._hooks = {}

    def on_read(self, key, hook):
        self._hooks[key] = hook

    def __getitem__(self, key):
        if self._hooks:
            h = self._hooks.get(key)
            if h is not None:
                h(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        if key in self:
            return self[key]
        return default
