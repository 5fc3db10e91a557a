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


class LeanStepRefused(RuntimeError):
    """Raised by a read hook for a key the lean step (args={"lean_step": True}) does not maintain and cannot reconstruct."""


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

    # every other way of getting at the values goes through the hooks too: values() / items() / copy(), and -- because __iter__
    # is overridden, CPython takes the keys() + __getitem__ path instead of copying the hash table -- dict(g) and **g
    def __iter__(self):
        return dict.__iter__(self)

    def _bulk(self, key):
        """bulk access (values / items / copy): a key whose hook REFUSES (the lean step's action history) is handed out as it is
        stored -- NaN-poisoned by whoever installed the hook -- instead of failing the whole iteration"""
        try:
            return self[key]
        except LeanStepRefused:  # only the refusal: a HIP error raised by a refresh hook propagates
            return dict.__getitem__(self, key)

    def values(self):
        return [self._bulk(k) for k in dict.keys(self)]

    def items(self):
        return [(k, self._bulk(k)) for k in dict.keys(self)]

    def copy(self):
        return {k: self._bulk(k) for k in dict.keys(self)}
