"""Golden vectors for the obstacle re-placement of a reset (row a20): the reference's AssetManager.reset_idx
(asset_manager.py:51-71) driven exactly as EnvManager.reset_idx drives it (env_manager.py:280-295: full reset of the
reset envs, then a second reset with half the obstacles for a bernoulli(0.15) subset).  Plain torch code: it RUNS.

    python oracle/gen_golden_assets.py        (in the build container: needs /root/reference)
"""
import os
import sys

import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402  FIRST: it switches TorchScript off before torch is imported (cr_torch.py)

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shells  # noqa: E402

OUT = gg.OUT


def main():
    os.makedirs(OUT, exist_ok=True)
    import types

    ref_shells.install()
    shell = types.ModuleType("aerial_gym.env_manager")  # the package's __init__ imports isaacgym: bypass it
    shell.__path__ = [os.path.join(ref_shells.REFERENCE_ROOT, "aerial_gym", "env_manager")]
    sys.modules.setdefault("aerial_gym.env_manager", shell)
    AssetManager = ref_shells.ref("env_manager.asset_manager").AssetManager
    g = torch.Generator().manual_seed(31)
    n, K = 40, 9
    lo = torch.rand(n, K, 13, generator=g) * 0.4
    hi = lo + torch.rand(n, K, 13, generator=g) * 0.5
    lo[..., 3:6] = (torch.rand(n, K, 3, generator=g) - 0.5) * 6.0  # euler ranges in radians
    hi[..., 3:6] = lo[..., 3:6] + torch.rand(n, K, 3, generator=g) * 2.0
    bmin = -torch.rand(n, 3, generator=g) * 6.0 - 1.0
    bmax = torch.rand(n, 3, generator=g) * 6.0 + 1.0
    state0 = torch.randn(n, K, 13, generator=g)
    out = {"min_ratio": lo.numpy(), "max_ratio": hi.numpy(), "bounds_min": bmin.numpy(), "bounds_max": bmax.numpy(),
           "state_before": state0.numpy()}
    for tag, num_obstacles, num_keep, p_env in (("a", 6, 2, 0.5), ("b", 1, 4, 0.8), ("c", 9, 0, 0.3)):
        gtd = {"env_asset_state_tensor": state0.clone(), "asset_min_state_ratio": lo, "asset_max_state_ratio": hi,
               "env_bounds_min": bmin, "env_bounds_max": bmax}
        am = AssetManager(gtd, num_keep)
        env_ids = torch.nonzero(torch.rand(n, generator=g) < p_env).squeeze(-1)
        # ---- env_manager.py:280-295, verbatim control flow
        torch.manual_seed(900 + num_obstacles)
        am.reset_idx(env_ids, num_obstacles)
        nk = am.num_keep_in_env
        am.num_keep_in_env = am.num_keep_in_env // 2
        samples = torch.bernoulli(0.15 * torch.ones(len(env_ids)))
        selected_indices = torch.nonzero(samples).squeeze(-1)
        rng_before_second = torch.get_rng_state()
        if len(selected_indices) > 0:
            am.reset_idx(env_ids[selected_indices], num_obstacles // 2)
        am.num_keep_in_env = nk
        # ---- replay of the draws (torch_rand_float_tensor is TorchScript: aten::rand_like on the global generator)
        torch.manual_seed(900 + num_obstacles)
        u1 = torch.rand(n, K, 13)
        torch.set_rng_state(rng_before_second)
        u2 = torch.rand(n, K, 13)
        mask = torch.zeros(n, dtype=torch.uint8)
        mask[env_ids] = 1
        sel = torch.zeros(n, dtype=torch.uint8)
        sel[env_ids[selected_indices]] = 1
        out.update({f"{tag}_params": np.array([num_obstacles, num_keep], np.int64), f"{tag}_mask": mask.numpy(), f"{tag}_sel": sel.numpy(),
                    f"{tag}_u1": u1.numpy(), f"{tag}_u2": u2.numpy(), f"{tag}_state_after": gtd["env_asset_state_tensor"].numpy()})
        print("asset_reset", tag, "reset envs", int(mask.sum()), "half-resampled", int(sel.sum()))
    np.savez_compressed(os.path.join(OUT, "asset_reset.npz"), **out)


if __name__ == "__main__":
    main()
