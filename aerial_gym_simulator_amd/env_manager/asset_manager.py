"""Obstacle pose randomisation at reset (aerial_gym/env_manager/asset_manager.py:51-71 and the
half-obstacle resample of env_manager.py:283-295) in agx_reset_assets, followed by the scene
rebuild of the reset envs (warp_env_manager.py:40-54): triangles, LBVH, collision OBBs."""

import torch

from .. import _lib


class AssetManager:
    def __init__(self, global_tensor_dict, num_keep_in_env, scene):
        self.g, self.scene = global_tensor_dict, scene
        self.num_keep_in_env = num_keep_in_env
        self.env_asset_state_tensor = global_tensor_dict["env_asset_state_tensor"]
        self.asset_min_state_ratio = global_tensor_dict["asset_min_state_ratio"]
        self.asset_max_state_ratio = global_tensor_dict["asset_max_state_ratio"]
        self._u1 = self._u2 = self._sel = None
        self.env = None

    def prepare_for_sim(self, env):
        self.env = env
        if self.scene.num_assets == 0 or env._buffers is None:
            return
        N, K, dev = self.scene.num_envs, self.scene.num_assets, self.scene.device
        if env.strict_rng:
            self._u1 = torch.zeros(N, K, 13, device=dev)
            self._u2 = torch.zeros(N, K, 13, device=dev)
            self._sel = torch.zeros(N, device=dev)
        # AssetManager.prepare_for_sim -> reset(num_keep_in_env) for every env (asset_manager.py:32-34)
        g = self.g
        g["reset_mask"].fill_(1)
        g["reset_flag"][env._parity] = 1
        if env.strict_rng:
            env.random_source.rand_into(env._u["bounds_lo"], tag="bounds_lo_init")
            env.random_source.rand_into(env._u["bounds_hi"], tag="bounds_hi_init")
            env.random_source.rand_into(self._u1, tag="assets_init")
            self._sel.zero_()
        self._apply(env, num_obstacles=self.num_keep_in_env)
        g["reset_mask"].zero_()
        g["reset_flag"].zero_()

    def draw_reset_randoms(self, env_ids):
        """strict_rng order (env_manager.py:283-295): rand_like [N,K,13]; bernoulli(0.15) over the
        reset envs; if any selected a second rand_like [N,K,13]."""
        if self.scene.num_assets == 0 or int(self.g["num_obstacles_in_env"]) == 0:
            return
        rs = self.env.random_source
        rs.rand_into(self._u1, tag="assets")
        self._sel.zero_()
        samples = rs.bernoulli(0.15, len(env_ids), tag="assets_half")
        self._sel[env_ids] = samples
        if bool((samples > 0).any()):
            rs.rand_into(self._u2, tag="assets_half_draw")

    def apply_env_actions(self, env, twists, k_substeps):
        """Kinematic obstacles: integrate the poses by k sub-steps of the commanded twists and move the
        geometry (triangles, BVH, collision boxes) of every env."""
        sc = self.scene
        N, K = sc.num_envs, sc.num_assets
        if K == 0:
            return
        tw = twists
        if tw.dtype != torch.float32 or not tw.is_contiguous():
            tw = tw.to(dtype=torch.float32).contiguous()
        if tw.shape != (N, K, 6):
            raise ValueError(f"env_actions must have shape ({N}, {K}, 6), got {tuple(tw.shape)}")
        lib, stream, p = env._lib, env._stream(), _lib.dptr
        st = self.env_asset_state_tensor
        _lib.check(lib.agx_assets_integrate(N, K, p(st), p(tw), float(self.g["dt"]), int(k_substeps), stream), "agx_assets_integrate")
        self._refresh_geometry(env, None)

    def reset_masked(self, env):
        # no assets, or none of them in the env at this curriculum level: the reference skips the asset reset
        # (env_manager.py:283-284: `if num_obstacles > 0:`), its draws included
        if self.scene.num_assets == 0 or int(self.g["num_obstacles_in_env"]) == 0:
            return
        self._apply(env, num_obstacles=int(self.g["num_obstacles_in_env"]))

    def _apply(self, env, num_obstacles):
        g, sc = self.g, self.scene
        N, K = sc.num_envs, sc.num_assets
        lib, stream, p = env._lib, env._stream(), _lib.dptr
        st = self.env_asset_state_tensor
        if not env.strict_rng and not sc.has_prims and env.env_args.get("fused_asset_reset", True):
            # obstacle poses, world-frame triangles, collision boxes and tree of the envs that reset: one call -- up to 2048 envs ONE
            # launch (the asset reset and the mask compaction are gone from the step), above, the three launches below
            _lib.check(
                lib.agx_scene_reset_refresh(env._buffers, N, sc.num_tris, K, env._reset_args, p(self.asset_min_state_ratio),
                                            p(self.asset_max_state_ratio), int(num_obstacles), int(self.num_keep_in_env), p(st),
                                            p(sc.tri_local), p(sc.tri_asset), p(sc.half_extents),
                                            int(getattr(env, 'bvh_prims_per_object', None) or sc.bvh_prims_per_object), p(sc.tri_world),
                                            p(sc.boxes_soa), p(sc.bvh_nodes), p(sc.bvh_work), stream),
                "agx_scene_reset_refresh",
            )
            return
        u1 = p(self._u1) if env.strict_rng else None
        u2 = p(self._u2) if env.strict_rng else None
        us = p(self._sel) if env.strict_rng else None
        _lib.check(
            lib.agx_reset_assets(env._buffers, N, K, env._reset_args, u1, u2, us, p(self.asset_min_state_ratio),
                                 p(self.asset_max_state_ratio), int(num_obstacles), int(self.num_keep_in_env), p(st), stream),
            "agx_reset_assets",
        )
        self._refresh_geometry(env, p(g["reset_mask"]))

    def _refresh_geometry(self, env, mk):
        """World-frame triangles, BVH and collision boxes of the envs flagged in `mk` (None = all); the kernels
        return immediately for clean envs.  Multi-primitive scenes first derive the primitive poses."""
        sc = self.scene
        N, K = sc.num_envs, sc.num_assets
        lib, stream, p = env._lib, env._stream(), _lib.dptr
        st, KP = self.env_asset_state_tensor, sc.num_prims
        if sc.has_prims:
            _lib.check(lib.agx_prims_from_assets(N, KP, K, p(sc.prim_asset), p(st), p(sc.prim_local_pos), p(sc.prim_local_quat), mk,
                                                 p(sc.prim_state), stream), "agx_prims_from_assets")
            st = sc.prim_state
        # triangles -> world frame, collision boxes, LBVH: one call (masked: one persistent launch over the dirty envs)
        _lib.check(lib.agx_scene_refresh(N, sc.num_tris, KP, p(sc.tri_local), p(sc.tri_asset), p(st), p(sc.half_extents),
                                         int(getattr(env, 'bvh_prims_per_object', None) or sc.bvh_prims_per_object), mk, p(sc.tri_world), p(sc.boxes_soa),
                                         p(sc.bvh_nodes), p(sc.bvh_work), stream), "agx_scene_refresh")
