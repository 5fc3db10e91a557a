"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Imports the *reference's own* PyTorch code (read-only, /root/reference) in THIS
container so that (a) the C oracle can be pinned against it and (b) golden
vectors can be generated (oracle/gen_golden.py).  /root/reference does not
exist on the GPU box, so nothing that runs there may import this module.

Why shells are needed: ``import aerial_gym`` fails because
aerial_gym/__init__.py:2 imports the closed-source ``isaacgym`` binary and
aerial_gym/utils/__init__.py:1 -> helpers.py:31 imports ``isaacgym.gymapi``.
We pre-register empty package objects for ``aerial_gym`` and
``aerial_gym.utils`` (with the right ``__path__``) so their sub-modules import
unmodified, and provide a stand-in for the one pytorch3d function on the hot
path (``matrix_to_quaternion``; pytorch3d is un-pinned in the reference's
setup.py:17 and not installed here).  The stand-in restates pytorch3d's
published argmax-branch algorithm (pytorch3d/transforms/rotation_conversions.py,
0.7.x) -- "parity unpinned" for that single function; every consumer on the
hot path is even in q, so the sign convention does not matter.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("AERIAL_GYM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "aerial_gym"))


def _matrix_to_quaternion(matrix):
    """pytorch3d.transforms.matrix_to_quaternion (returns wxyz), no sign standardisation."""
    import torch
    import torch.nn.functional as F

    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(
        matrix.reshape(batch_dim + (9,)), dim=-1
    )

    def _sqrt_positive_part(x):
        ret = torch.zeros_like(x)
        positive_mask = x > 0
        ret[positive_mask] = torch.sqrt(x[positive_mask])
        return ret

    q_abs = _sqrt_positive_part(
        torch.stack(
            [
                1.0 + m00 + m11 + m22,
                1.0 + m00 - m11 - m22,
                1.0 - m00 + m11 - m22,
                1.0 - m00 - m11 + m22,
            ],
            dim=-1,
        )
    )
    quat_by_rijk = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    return quat_candidates[
        F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :
    ].reshape(batch_dim + (4,))


def install():
    """Register the shells; idempotent.  Returns the ``aerial_gym`` shell module."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    if "aerial_gym" in sys.modules and getattr(sys.modules["aerial_gym"], "_IS_SHELL", False):
        return sys.modules["aerial_gym"]
    pkg_dir = os.path.join(REFERENCE_ROOT, "aerial_gym")

    ag = types.ModuleType("aerial_gym")
    ag.__path__ = [pkg_dir]
    ag.AERIAL_GYM_DIRECTORY = REFERENCE_ROOT
    ag._IS_SHELL = True
    sys.modules["aerial_gym"] = ag

    utils = types.ModuleType("aerial_gym.utils")
    utils.__path__ = [os.path.join(pkg_dir, "utils")]
    sys.modules["aerial_gym.utils"] = utils
    ag.utils = utils

    if "pytorch3d" not in sys.modules:
        p3d = types.ModuleType("pytorch3d")
        p3d.__path__ = []
        p3d_t = types.ModuleType("pytorch3d.transforms")
        p3d_t.matrix_to_quaternion = _matrix_to_quaternion
        p3d.transforms = p3d_t
        sys.modules["pytorch3d"] = p3d
        sys.modules["pytorch3d.transforms"] = p3d_t
    return ag


def ref(module: str):
    """Import a reference sub-module, e.g. ref("utils.math")."""
    install()
    return importlib.import_module("aerial_gym." + module)


def install_task_shells():
    """Extra shells so the reference's TASK modules import (for their jit reward functions).

    position_setpoint_task.py:1-11 / navigation_task.py:1-13 import SimBuilder (-> isaacgym),
    gymnasium, gym.spaces and the VAE encoder; none is needed to call ``compute_reward``.
    """
    install()
    import torch  # noqa: F401

    def _mod(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Space:
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k

    _mod("gymnasium", spaces=_mod("gymnasium.spaces", Box=_Space, Dict=_Space))
    _mod("gym", spaces=_mod("gym.spaces", Box=_Space, Dict=_Space))
    _mod("aerial_gym.sim")
    _mod("aerial_gym.sim.sim_builder", SimBuilder=object)
    _mod("aerial_gym.utils.vae")
    _mod("aerial_gym.utils.vae.vae_image_encoder", VAEImageEncoder=object)
    # task/__init__.py registers every task (imports half the tree): bypass it
    task_pkg = types.ModuleType("aerial_gym.task")
    task_pkg.__path__ = [os.path.join(REFERENCE_ROOT, "aerial_gym", "task")]
    sys.modules.setdefault("aerial_gym.task", task_pkg)
    for sub in ("position_setpoint_task", "navigation_task"):
        sp = types.ModuleType("aerial_gym.task." + sub)
        sp.__path__ = [os.path.join(REFERENCE_ROOT, "aerial_gym", "task", sub)]
        sys.modules.setdefault("aerial_gym.task." + sub, sp)
