"""Obstacle pose randomisation at reset (aerial_gym/env_manager/asset_manager.py:51-71 and
the half-obstacle resample of env_manager.py:283-295), followed by the scene rebuild of the
reset envs (warp_env_manager.py:40-54)."""
import torch

from .. import _lib
from ..utils.math import quat_from_euler_xyz_tensor


class AssetManager:
    def __init__(self, global_tensor_dict, num_keep_in_env, scene):
        self.g, self.scene = global_tensor_dict, scene
        self.num_keep_in_env = num_keep_in_env
        self.env_asset_state_tensor = global_tensor_dict["env_asset_state_tensor"]
        self.asset_min_state_ratio = global_tensor_dict["asset_min_state_ratio"]
        self.asset_max_state_ratio = global_tensor_dict["asset_max_state_ratio"]
        self._u1 = self._u2 = self._sel = None

    def prepare_for_sim(self, env):
        self.env = env
        if self.scene.num_assets == 0 or env._buffers is None:
            return
        N, K, dev = self.scene.num_envs, self.scene.num_assets, self.scene.device
        self._u1 = torch.zeros(N, K, 13, device=dev)
        self._u2 = torch.zeros(N, K, 13, device=dev)
        self._sel = torch.zeros(N, device=dev)
        # AssetManager.prepare_for_sim -> reset(num_keep_in_env) for every env (asset_manager.py:32-34)
        g = self.g
        g["reset_mask"].fill_(1)
        g["reset_flag"].fill_(1)
        env.random_source.rand_into(self._u1, tag="assets_init")
        self._sel.zero_()
        self._apply(env, num_obstacles=self.num_keep_in_env)
        g["reset_mask"].zero_()
        g["reset_flag"].zero_()

    def draw_reset_randoms(self, env_ids):
        """strict_rng order (env_manager.py:283-295): rand_like [N,K,13]; bernoulli(0.15) over the
        reset envs; if any selected a second rand_like [N,K,13]."""
        if self.scene.num_assets == 0:
            return
        rs = self.env.random_source
        rs.rand_into(self._u1, tag="assets")
        self._sel.zero_()
        samples = rs.bernoulli(0.15, len(env_ids), tag="assets_half")
        self._sel[env_ids] = samples
        if bool((samples > 0).any()):
            rs.rand_into(self._u2, tag="assets_half_draw")

    def reset_masked(self, env):
        if self.scene.num_assets == 0:
            return
        if not env.strict_rng:
            raise NotImplementedError("sync-free obstacle reset needs the device RNG path (see DESIGN.md)")
        self._apply(env, num_obstacles=int(self.g["num_obstacles_in_env"]))

    def _apply(self, env, num_obstacles):
        """Host-side (torch, on device) evaluation of the two-phase asset reset; runs only on
        steps where some env resets.  Then the HIP scene kernels rebuild triangles/BVH/OBBs."""
        g, sc = self.g, self.scene
        N, K = sc.num_envs, sc.num_assets
        mask = g["reset_mask"].bool()
        nk = self.num_keep_in_env
        n_full = max(num_obstacles, nk)
        n_half = max(num_obstacles // 2, nk // 2)
        sel = (self._sel > 0) & mask
        u = torch.where(sel.view(N, 1, 1), self._u2, self._u1)
        ratio = (self.asset_max_state_ratio - self.asset_min_state_ratio) * u + self.asset_min_state_ratio
        bmin = g["env_bounds_min"].unsqueeze(1)
        bmax = g["env_bounds_max"].unsqueeze(1)
        pos = bmin + (bmax - bmin) * ratio[..., 0:3]
        quat = quat_from_euler_xyz_tensor(ratio[..., 3:6])
        n_active = torch.where(sel, n_half, n_full).view(N, 1)
        parked = torch.arange(K, device=sc.device).view(1, K) >= n_active
        pos = torch.where(parked.unsqueeze(-1), torch.full_like(pos, -1000.0), pos)
        m3 = mask.view(N, 1, 1)
        st = self.env_asset_state_tensor
        st[..., 0:3] = torch.where(m3, pos, st[..., 0:3])
        st[..., 3:7] = torch.where(m3, quat, st[..., 3:7])
        lib, stream, p = env._lib, env._stream(), _lib.dptr
        mk = p(g["reset_mask"])
        _lib.check(lib.agx_scene_transform(N, sc.num_tris, K, p(sc.tri_local), p(sc.tri_asset), p(st), mk, p(sc.tri_world), stream),
                   "agx_scene_transform")
        _lib.check(lib.agx_bvh_build(N, sc.num_tris, p(sc.tri_world), mk, p(sc.bvh_nodes), stream), "agx_bvh_build")
        _lib.check(lib.agx_boxes_from_assets(N, K, p(st), p(sc.half_extents), mk, p(sc.boxes_soa), stream), "agx_boxes_from_assets")
