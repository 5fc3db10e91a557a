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
    def __init__(self, env_cfg, num_envs, device, random_source, shard_rank=0, scene_seed_base=1000, env_offset=None, box_objects=True):
        self.cfg, self.num_envs, self.device = env_cfg, num_envs, device
        self.shard_rank = shard_rank
        # this process owns global envs [env_offset, env_offset + N)  (SURVEY 8e); equal shards: rank * N
        self.env_offset = env_offset = shard_rank * num_envs if env_offset is None else int(env_offset)
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
        self.num_prims, self.has_prims = K, False  # collision / scene pieces: one per asset unless a URDF has several links
        # objects of 12 consecutive triangles (agx_bvh_build) + AGX_BVH_BOX_OBJECTS: where an object's 12 triangles ARE a box of
        # trimesh's topology -- the builder checks every vertex -- the tree ends at the object and the ray-cast kernels intersect
        # the box's frame, running the exact triangle test on the entered face only (same frames, bit for bit); chunks of a
        # cylinder / sphere / mesh keep their triangle subtrees.  box_objects=False (EnvManager args={"bvh_box_objects": False}):
        # triangle subtrees everywhere (A/B runs).
        self._box_objects = bool(box_objects)
        # sharding: start of this rank's slice of the global asset counter
        semantic_offset = self.semantic_offset = env_offset * K  # this shard's slice of the global id counter
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
        variants = [self._variants(t) for t in slots]
        if any(len(v) != 1 or v[0].kind != "box" or not np.allclose(v[0].T, np.eye(4)) for vs in variants for v in vs):
            self._init_general(slots, variants, nk, nf, N, scene_seed_base, env_offset)
            return
        choices = [[v[0].dims for v in vs] for vs in variants]
        max_choices = max(len(c) for c in choices)
        slot_nchoice = np.array([len(c) for c in choices])
        slot_choices = np.zeros((len(slots), max_choices, 3), np.float32)
        for j, c in enumerate(choices):
            slot_choices[j, : len(c)] = c
        free_idx = list(range(nk, nk + nf))
        for i in range(N):
            rng = np.random.default_rng(scene_seed_base + env_offset + i)  # seeded by GLOBAL env index
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
    def _variants(cls, acfg):
        """The shapes an instance of this asset type may take: a list of variants, each a list of primitives
        (assets.Prim).  One variant is drawn per instance and env (asset_loader.py:44-56)."""
        from ..assets import Prim, list_urdf_files, load_urdf_primitives

        folder = getattr(acfg, "asset_folder", None)
        if folder and os.path.isdir(folder):
            files = [acfg.file] if getattr(acfg, "file", None) else list_urdf_files(folder)
            if not files:
                raise ValueError(f"no URDF files in {folder}")
            key = (folder, tuple(files), bool(getattr(acfg, "use_collision_mesh_instead_of_visual", False)))
            if key not in cls._urdf_cache:  # asset_loader.py:66-70 keeps a buffer of loaded files, too
                cls._urdf_cache[key] = [load_urdf_primitives(os.path.join(folder, f), key[2]) for f in files]
            return cls._urdf_cache[key]
        if folder and getattr(acfg, "box_sizes", None) is None and getattr(acfg, "random_box_size_range", None) is None:
            raise FileNotFoundError(f"asset folder {folder} does not exist (point asset_folder at your aerial_gym "
                                    "resources/models/environment_assets/<set> directory)")
        sizes = list(acfg.box_sizes) if acfg.box_sizes else [[0.0, 0.0, 0.0]]
        return [[Prim("box", tuple(float(v) for v in sz), np.eye(4), "base_link", 0)] for sz in sizes]

    @property
    def bvh_prims_per_object(self):
        """agx_bvh_build's prims_per_object: chunks of 12 triangles | AGX_BVH_BOX_OBJECTS | -- where every object IS a box (no URDF
        with curved / mesh links in the scene) -- AGX_BVH_OBJECT_TREE: the tree is built over the objects (csrc/agx_scene.hip).  The
        quick subtrees that build gives non-box chunks traverse 7-9 % slower than their LBVH subtrees (profiles/forest_probe_r06.py):
        scenes with primitives keep the triangle-level build, object nodes for their boxes included."""
        if not self._box_objects:
            return 12
        return 12 | 0x20000000 | (0 if self.has_prims else 0x10000000)

    MAX_TRIS_PER_ENV = 2944  # csrc/agx_scene.hip kBvhMaxTris (agx_bvh_build refuses more by message; this names the assets)

    def _init_general(self, slots, variants, nk, nf, N, scene_seed_base, env_offset):
        """Scenes with multi-primitive assets (several links, cylinders): every primitive is its own rigid piece
        (own triangles in its own frame, own collision box) tied to its asset by agx_prims_from_assets.  The
        primitive layout is fixed by the canonical slot order; which asset INDEX owns a slot differs per env
        (the reference shuffles the free assets of every env, asset_loader.py:179-183)."""
        from ..assets import half_extents, num_triangles, quat_xyzw_from_matrix, tessellate

        K = len(slots)
        self.has_prims = True
        P_s = [max(len(v) for v in vs) for vs in variants]                    # primitives per slot
        tri_s = [[max(num_triangles(v[q] if q < len(v) else v[0]) for v in vs) for q in range(P_s[s])]
                 for s, vs in enumerate(variants)]                            # triangles per primitive slot
        prim_base = np.concatenate([[0], np.cumsum(P_s)]).astype(int)
        KP = self.num_prims = int(prim_base[-1])
        tri_count = np.array([t for ts in tri_s for t in ts], int)
        tri_base = np.concatenate([[0], np.cumsum(tri_count)]).astype(int)
        T = self.num_tris = int(tri_base[-1])
        if T > self.MAX_TRIS_PER_ENV:
            per_slot = sorted(((sum(ts), getattr(slots[s], "__name__", type(slots[s]).__name__)) for s, ts in enumerate(tri_s)), reverse=True)[:5]
            raise ValueError(
                f"scene of {T} triangles per env exceeds the {self.MAX_TRIS_PER_ENV} the LDS-resident tree build takes "
                f"(csrc/agx_scene.hip kBvhMaxTris); largest assets (triangles, type): {per_slot}.  A URDF sphere is 1280 triangles "
                "(trimesh icosphere, 3 subdivisions) and a cylinder 128 (32 sections): use fewer of them, or box primitives")
        ids_per_slot = np.array([max(len(v) for v in vs) if getattr(t, "per_link_semantic", False) and t.semantic_id < 0 else 1
                                 for t, vs in zip(slots, variants)])
        semantic_offset = self.semantic_offset = env_offset * int(ids_per_slot.sum())
        lo, hi = np.zeros((N, K, 13), np.float32), np.zeros((N, K, 13), np.float32)
        prim_asset = np.zeros((N, KP), np.int32)
        prim_half, prim_lpos = np.zeros((N, KP, 3), np.float32), np.zeros((N, KP, 3), np.float32)
        prim_lquat = np.zeros((N, KP, 4), np.float32)
        prim_sem = np.zeros((N, KP), np.int64)
        tri_local = np.zeros((N, T, 9), np.float32)
        asset_sem = np.zeros((N, K), np.int64)
        counter = 100 + semantic_offset
        free_idx = list(range(nk, nk + nf))
        for i in range(N):
            rng = np.random.default_rng(scene_seed_base + env_offset + i)
            order = free_idx[:]
            random.shuffle(order)
            perm = list(range(nk)) + order
            u = rng.uniform(0.0, 1.0, (K, 3)).astype(np.float32)
            pick = rng.integers(0, 1 << 30, K)
            for j, s_ in enumerate(perm):  # asset index j of this env is canonical slot s_
                t = slots[s_]
                lo[i, j], hi[i, j] = t.min_state_ratio, t.max_state_ratio
                vs = variants[s_]
                v = vs[int(pick[j]) % len(vs)]
                if getattr(t, "random_box_size_range", None) is not None:
                    rl, rh = np.array(t.random_box_size_range[0], np.float32), np.array(t.random_box_size_range[1], np.float32)
                    from ..assets import Prim
                    v = [Prim("box", tuple(float(x) for x in (rl + (rh - rl) * u[j])), np.eye(4), "base_link", 0)]
                base_id = t.semantic_id if t.semantic_id >= 0 else counter
                asset_sem[i, j] = base_id
                links = sorted({p.link_index for p in v})
                for q in range(P_s[s_]):
                    pr = v[q] if q < len(v) else v[0]  # padding = a duplicate of primitive 0 (same hits, same ids)
                    pg = prim_base[s_] + q
                    prim_asset[i, pg] = j
                    prim_half[i, pg] = half_extents(pr)
                    prim_lpos[i, pg] = pr.T[:3, 3]
                    prim_lquat[i, pg] = quat_xyzw_from_matrix(pr.T[:3, :3])
                    per_link = getattr(t, "per_link_semantic", False) and t.semantic_id < 0
                    prim_sem[i, pg] = base_id + (links.index(pr.link_index) if per_link else 0)
                    tr = tessellate(pr).reshape(-1, 9)
                    tri_local[i, tri_base[pg]: tri_base[pg] + len(tr)] = tr
                    if len(tr) < tri_count[pg]:  # a box in a slot that may also hold a cylinder / a smaller mesh: repeat triangles
                        if tri_count[pg] % len(tr) == 0:
                            tri_local[i, tri_base[pg]: tri_base[pg + 1]] = np.tile(tr, (tri_count[pg] // len(tr), 1))
                        else:
                            tri_local[i, tri_base[pg] + len(tr): tri_base[pg + 1]] = tr[-1]
                if t.semantic_id < 0:
                    counter += int(ids_per_slot[s_])
        tri_prim = np.repeat(np.arange(KP, dtype=np.int32), tri_count)
        self._np = dict(lo=lo, hi=hi, sem=asset_sem, size=np.zeros((N, K, 3), np.float32), prim_asset=prim_asset, prim_half=prim_half,
                        prim_lpos=prim_lpos, prim_lquat=prim_lquat, prim_sem=prim_sem, tri_local=tri_local, tri_prim=tri_prim)

    def prepare_for_simulation(self, global_tensor_dict):
        g, N, dev, K = global_tensor_dict, self.num_envs, self.device, self.num_assets
        if K == 0:
            g["env_asset_state_tensor"] = torch.zeros(N, 0, 13, device=dev)
            g["asset_min_state_ratio"] = torch.zeros(N, 0, 13, device=dev)
            g["asset_max_state_ratio"] = torch.zeros(N, 0, 13, device=dev)
            self.boxes_soa = None
            return
        d = self._np
        if self.has_prims:
            return self._prepare_general(g, d)
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

    def _prepare_general(self, g, d):
        N, dev, K, KP, T = self.num_envs, self.device, self.num_assets, self.num_prims, self.num_tris
        t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)  # noqa: E731
        g["asset_min_state_ratio"], g["asset_max_state_ratio"] = t(d["lo"]), t(d["hi"])
        self.asset_semantic_id = t(d["sem"])
        self.asset_state = torch.zeros(N, K, 13, device=dev)
        self.asset_state[..., 6] = 1.0
        g["env_asset_state_tensor"] = self.asset_state
        g["obstacle_position"], g["obstacle_orientation"] = self.asset_state[..., 0:3], self.asset_state[..., 3:7]
        g["obstacle_linvel"], g["obstacle_angvel"] = self.asset_state[..., 7:10], self.asset_state[..., 10:13]
        # primitives: the pieces the scene kernels see in place of assets
        self.prim_asset = t(d["prim_asset"])            # [N, P] owning asset index (per env: the free assets are shuffled)
        self.prim_local_pos, self.prim_local_quat = t(d["prim_lpos"]), t(d["prim_lquat"])
        self.half_extents = t(d["prim_half"])           # [N, P, 3]
        self.prim_state = torch.zeros(N, KP, 13, device=dev)
        self.prim_state[..., 6] = 1.0
        self.tri_local = t(d["tri_local"])
        self.tri_world = torch.zeros_like(self.tri_local)
        self.tri_asset = t(d["tri_prim"])               # triangle -> primitive index
        self.tri_seg = t(d["prim_sem"]).to(torch.int32)[:, self.tri_asset.long()].contiguous()
        self.boxes_soa = torch.zeros(KP * 11, N, device=dev)
        self.bvh_nodes = torch.zeros(N, max(T - 1, 1), 16, device=dev)
        self.bvh_work = torch.zeros(N + 2, dtype=torch.int32, device=dev)
        g["scene_tri_world"], g["scene_tri_seg"], g["scene_bvh_nodes"] = self.tri_world, self.tri_seg, self.bvh_nodes
