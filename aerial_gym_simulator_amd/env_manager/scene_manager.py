"""Per-env obstacle scenes: what WarpEnv + AssetLoader + IsaacGymEnv.add_asset_to_env build
in the reference (warp_env_manager.py:74-189, asset_loader.py:148-194, env_manager.py:147-232),
for box obstacles, vectorised over envs (the reference loops env x asset in Python).

Device data owned here
    asset_state   [N, K, 13]  env_asset_state_tensor (reference layout)
    half_extents  [N, K, 3]
    tri_local     [N, T, 9]   box triangles in the asset frame (T = 12 K, trimesh box topology)
    tri_world     [N, T, 9]   after tf_apply(asset pose)   (rebuilt for reset envs)
    tri_seg       [N, T]      int32 segmentation id of the owning asset
    bvh_nodes     [N, T-1, 16] LBVH built on device by agx_bvh_build
    boxes_soa     [K*11, N]   OBBs (+ bounding radius) for the collision test
"""
import os
import random

import numpy as np
import torch

from ..sharding import semantic_id_offset

# trimesh.creation.box: vertices = ({0,1}^3 - 0.5) * extents in this order, 12 faces
_BOX_VERTS = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]], dtype=np.float32) - 0.5
_BOX_FACES = np.array(
    [[1, 3, 0], [4, 1, 0], [0, 3, 2], [2, 4, 0], [1, 7, 3], [5, 1, 4], [5, 7, 1], [3, 7, 2], [6, 4, 2], [2, 7, 6], [6, 5, 4], [7, 5, 6]],
    dtype=np.int64,
)


class SceneManager:
    def __init__(self, env_cfg, num_envs, device, random_source, shard_rank=0, scene_seed_base=1000):
        self.cfg, self.num_envs, self.device = env_cfg, num_envs, device
        self.shard_rank = shard_rank  # this process owns global envs [rank * N, (rank + 1) * N)  
        N = num_envs
        types = []
        for name, acfg in env_cfg.env_config.asset_type_to_dict_map.items():
            if env_cfg.env_config.include_asset_type.get(name, True) and acfg.num_assets > 0:
                types.append(acfg)
        keep = [t for t in types if t.keep_in_env]
        free = [t for t in types if not t.keep_in_env]
        # asset_loader.py:148-183: keep_in_env assets are appendleft'ed (reverse insertion order),
        # the rest appended then shuffled per env
        keep_slots = [t for t in keep for _ in range(t.num_assets)][::-1]
        free_slots = [t for t in free for _ in range(t.num_assets)]
        self.keep_in_env_num = len(keep_slots)
        K = self.num_assets = len(keep_slots) + len(free_slots)
        self.num_tris = 12 * K
        # sharding: start of this rank's slice of the global asset counter
        semantic_offset = self.semantic_offset = semantic_id_offset(shard_rank, N, K)
        if K == 0:
            return
        size = np.zeros((N, K, 3), np.float32)
        lo = np.zeros((N, K, 13), np.float32)
        hi = np.zeros((N, K, 13), np.float32)
        sem = np.zeros((N, K), np.int64)
        nk, nf = len(keep_slots), len(free_slots)
        # per-slot constant tables (slot = one asset instance of one type), gathered per env below
        slots = keep_slots + free_slots
        slot_lo = np.array([t.min_state_ratio for t in slots], np.float32)
        slot_hi = np.array([t.max_state_ratio for t in slots], np.float32)
        slot_sem = np.array([t.semantic_id for t in slots], np.int64)
        slot_random = np.array([t.random_box_size_range is not None for t in slots])
        slot_rlo = np.array([t.random_box_size_range[0] if t.random_box_size_range else [0, 0, 0] for t in slots], np.float32)
        slot_rhi = np.array([t.random_box_size_range[1] if t.random_box_size_range else [0, 0, 0] for t in slots], np.float32)
        # geometry per asset type: URDF folder when configured and present, else the restated box-size table
        choices = [self._box_choices(t) for t in slots]
        max_choices = max(len(c) for c in choices)
        slot_nchoice = np.array([len(c) for c in choices])
        slot_choices = np.zeros((len(slots), max_choices, 3), np.float32)
        for j, c in enumerate(choices):
            slot_choices[j, : len(c)] = c
        free_idx = list(range(nk, nk + nf))
        for i in range(N):
            rng = np.random.default_rng(scene_seed_base + shard_rank * N + i)  # seeded by GLOBAL env index
            order = free_idx[:]
            random.shuffle(order)  # python `random`, like asset_loader.py:181
            perm = np.array(list(range(nk)) + order)
            u = rng.uniform(0.0, 1.0, (K, 3)).astype(np.float32)
            pick = (rng.integers(0, 1 << 30, K) % slot_nchoice[perm])
            fixed = slot_choices[perm, pick]
            rand = slot_rlo[perm] + (slot_rhi[perm] - slot_rlo[perm]) * u
            size[i] = np.where(slot_random[perm][:, None], rand, fixed)
            lo[i], hi[i], sem[i] = slot_lo[perm], slot_hi[perm], slot_sem[perm]
        # env_manager.py:147,212 + warp_env_manager.py:74-95: global counter from 100, one per asset
        counter = 100 + semantic_offset + np.arange(N * K).reshape(N, K)
        sem = np.where(sem < 0, counter, sem)
        self._np = dict(size=size, lo=lo, hi=hi, sem=sem)

    _urdf_cache = {}

    @classmethod
    def _box_choices(cls, acfg):
        """Box sizes an instance of this asset type may take (one is drawn per instance and env)."""
        folder = getattr(acfg, "asset_folder", None)
        if folder and os.path.isdir(folder):
            from ..assets import list_urdf_files, parse_box_urdf

            files = [acfg.file] if getattr(acfg, "file", None) else list_urdf_files(folder)
            if not files:
                raise ValueError(f"no URDF files in {folder}")
            key = (folder, tuple(files), bool(getattr(acfg, "use_collision_mesh_instead_of_visual", False)))
            if key not in cls._urdf_cache:  # asset_loader.py:66-70 keeps a buffer of loaded files, too
                cls._urdf_cache[key] = [parse_box_urdf(os.path.join(folder, f), key[2]).size for f in files]
            return cls._urdf_cache[key]
        return list(acfg.box_sizes) if acfg.box_sizes else [[0.0, 0.0, 0.0]]

    def prepare_for_simulation(self, global_tensor_dict):
        g, N, dev, K = global_tensor_dict, self.num_envs, self.device, self.num_assets
        if K == 0:
            g["env_asset_state_tensor"] = torch.zeros(N, 0, 13, device=dev)
            g["asset_min_state_ratio"] = torch.zeros(N, 0, 13, device=dev)
            g["asset_max_state_ratio"] = torch.zeros(N, 0, 13, device=dev)
            self.boxes_soa = None
            return
        d = self._np
        self.half_extents = torch.from_numpy(d["size"] * 0.5).to(dev)
        g["asset_min_state_ratio"] = torch.from_numpy(d["lo"]).to(dev)
        g["asset_max_state_ratio"] = torch.from_numpy(d["hi"]).to(dev)
        self.asset_semantic_id = torch.from_numpy(d["sem"]).to(dev)
        self.asset_state = torch.zeros(N, K, 13, device=dev)
        self.asset_state[..., 6] = 1.0
        g["env_asset_state_tensor"] = self.asset_state
        g["obstacle_position"] = self.asset_state[..., 0:3]
        g["obstacle_orientation"] = self.asset_state[..., 3:7]
        g["obstacle_linvel"] = self.asset_state[..., 7:10]
        g["obstacle_angvel"] = self.asset_state[..., 10:13]
        verts = torch.from_numpy(_BOX_VERTS).to(dev)  # [8,3]
        faces = torch.from_numpy(_BOX_FACES).to(dev)  # [12,3]
        box_tris = verts[faces]  # [12,3,3]
        # [N,K,12,3,3] = unit-box triangle * size
        size = torch.from_numpy(d["size"]).to(dev)
        self.tri_local = (box_tris.view(1, 1, 12, 3, 3) * size.view(N, K, 1, 1, 3)).reshape(N, 12 * K, 9).contiguous()
        self.tri_world = torch.zeros_like(self.tri_local)
        self.tri_asset = torch.arange(K, device=dev, dtype=torch.int32).repeat_interleave(12).contiguous()
        self.tri_seg = self.asset_semantic_id.to(torch.int32).repeat_interleave(12, dim=1).contiguous()
        self.boxes_soa = torch.zeros(K * 11, N, device=dev)
        self.bvh_nodes = torch.zeros(N, max(12 * K - 1, 1), 16, device=dev)
        self.bvh_work = torch.zeros(N + 2, dtype=torch.int32, device=dev)  # dirty-env work list of agx_bvh_build
        g["scene_tri_world"] = self.tri_world
        g["scene_tri_seg"] = self.tri_seg
        g["scene_bvh_nodes"] = self.bvh_nodes
