"""CPU: box obstacles from URDF files (box subset) and the restated box-size tables
checked against the reference's own URDFs when the reference tree is present (build container)."""
import os

import numpy as np
import pytest

REF_ASSETS = "/root/reference/resources/models/environment_assets"

BOX_URDF = """<?xml version='1.0'?>
<robot name="box"><link name="base_link">
  <inertial><origin xyz="0 0 0" rpy="0 0 0"/><mass value="1.0"/><inertia ixx="1" ixy="0" ixz="0" iyy="1" iyz="0" izz="1"/></inertial>
  <visual name="v"><geometry><box size="{vis}"/></geometry><origin xyz="0 0 0" rpy="0 0 0"/></visual>
  <collision name="c"><geometry><box size="{col}"/></geometry><origin xyz="0 0 0" rpy="0 0 0"/></collision>
</link></robot>"""


def _write(tmp_path, name, vis, col=None):
    (tmp_path / name).write_text(BOX_URDF.format(vis=vis, col=col or vis))


def test_parse_box_urdf_and_rejections(tmp_path):
    from aerial_gym_simulator_amd.assets import list_urdf_files, parse_box_urdf

    _write(tmp_path, "a.urdf", "0.1 0.5 0.5", "0.2 0.6 0.6")
    (tmp_path / "notes.txt").write_text("not an asset")
    a = parse_box_urdf(str(tmp_path / "a.urdf"))
    assert a.size == (0.1, 0.5, 0.5) and a.file == "a.urdf"
    assert parse_box_urdf(str(tmp_path / "a.urdf"), use_collision=True).size == (0.2, 0.6, 0.6)
    assert list_urdf_files(str(tmp_path)) == ["a.urdf"]
    (tmp_path / "cyl.urdf").write_text(BOX_URDF.format(vis="1 1 1", col="1 1 1").replace('<box size="1 1 1"/>', '<cylinder radius="0.1" length="2"/>'))
    with pytest.raises(NotImplementedError, match="not a box"):
        parse_box_urdf(str(tmp_path / "cyl.urdf"))
    two = BOX_URDF.format(vis="1 1 1", col="1 1 1").replace("</robot>", '<link name="l2"/></robot>')
    (tmp_path / "two.urdf").write_text(two)
    with pytest.raises(NotImplementedError, match="2 links"):
        parse_box_urdf(str(tmp_path / "two.urdf"))
    _write(tmp_path, "bad.urdf", "1 -1 1")
    with pytest.raises(ValueError, match="non-positive"):
        parse_box_urdf(str(tmp_path / "bad.urdf"))


def test_scene_from_a_urdf_folder(tmp_path):
    """An asset type that points at a folder of URDFs (asset_folder / file = None -> random pick per instance)."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config import asset_config as A
    from aerial_gym_simulator_amd.config.env_config import EnvWithObstaclesCfg
    from aerial_gym_simulator_amd.env_manager.scene_manager import SceneManager

    _write(tmp_path, "small.urdf", "0.3 0.3 0.3")
    _write(tmp_path, "rod.urdf", "0.1 0.1 1.5")

    class folder_objects(A.object_asset_params):
        num_assets = 20
        asset_folder = str(tmp_path)
        file = None

    class fixed_file(A.object_asset_params):
        num_assets = 3
        asset_folder = str(tmp_path)
        file = "rod.urdf"

    class Cfg(EnvWithObstaclesCfg):
        class env_config:
            include_asset_type = {"a": True, "b": True}
            asset_type_to_dict_map = {"a": folder_objects, "b": fixed_file}

    sc = SceneManager(Cfg, 5, "cpu", None)
    assert sc.num_assets == 23 and sc.num_tris == 12 * 23
    size = sc._np["size"]
    kinds = {tuple(round(float(v), 4) for v in s) for s in size.reshape(-1, 3)}
    assert kinds == {(0.3, 0.3, 0.3), (0.1, 0.1, 1.5)}
    rods = (size == np.float32([0.1, 0.1, 1.5])).all(-1).sum(axis=1)
    assert rods.min() >= 3 and rods.max() < 23  # the three fixed rods + a random share of the folder picks


@pytest.mark.skipif(not os.path.isdir(REF_ASSETS), reason="reference tree not present (GPU box)")
def test_restated_box_tables_match_the_reference_urdfs():
    """config/asset_config.py restates the reference's obstacle URDFs as data: every shipped box file must be in
    the table of its asset type, and nothing else."""
    from aerial_gym_simulator_amd.assets import list_urdf_files, parse_box_urdf
    from aerial_gym_simulator_amd.config import asset_config as A

    def sizes(folder):
        return sorted(parse_box_urdf(os.path.join(REF_ASSETS, folder, f)).size for f in list_urdf_files(os.path.join(REF_ASSETS, folder)))

    assert sizes("objects") == sorted(tuple(float(v) for v in s) for s in A.object_asset_params.box_sizes)
    assert sizes("panels") == sorted(tuple(float(v) for v in s) for s in A.panel_asset_params.box_sizes)
    walls = {w.__name__: tuple(float(v) for v in w.box_sizes[0]) for w in (A.left_wall, A.right_wall, A.top_wall, A.bottom_wall, A.front_wall, A.back_wall)}
    for name, size in walls.items():
        assert parse_box_urdf(os.path.join(REF_ASSETS, "walls", name + ".urdf")).size == size, name
    # the same env built from the reference's folders and from the restated tables has the same box multiset
    from aerial_gym_simulator_amd.config.env_config import EnvWithObstaclesCfg
    from aerial_gym_simulator_amd.env_manager.scene_manager import SceneManager

    class ref_objects(A.object_asset_params):
        asset_folder = os.path.join(REF_ASSETS, "objects")

    class ref_panels(A.panel_asset_params):
        asset_folder, file = os.path.join(REF_ASSETS, "panels"), "panel.urdf"

    class Cfg(EnvWithObstaclesCfg):
        class env_config:
            include_asset_type = dict(EnvWithObstaclesCfg.env_config.include_asset_type)
            asset_type_to_dict_map = dict(EnvWithObstaclesCfg.env_config.asset_type_to_dict_map, panels=ref_panels, objects=ref_objects)

    a, b = SceneManager(Cfg, 3, "cpu", None), SceneManager(EnvWithObstaclesCfg, 3, "cpu", None)
    assert a.num_assets == b.num_assets == 44
    assert {tuple(s) for s in a._np["size"].reshape(-1, 3)} == {tuple(s) for s in b._np["size"].reshape(-1, 3)}
    # with the trees / thin sets (cylinders) the loader says what is missing instead of guessing
    with pytest.raises(NotImplementedError):
        parse_box_urdf(os.path.join(REF_ASSETS, "trees", "tree_0.urdf"))


FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "assets")


def test_urdf_primitives_forward_kinematics_and_tessellation():
    from aerial_gym_simulator_amd.assets import half_extents, load_urdf_primitives, quat_xyzw_from_matrix, tessellate
    from aerial_gym_simulator_amd.assets.urdf_primitives import rpy_matrix

    ps = load_urdf_primitives(os.path.join(FIX, "trees", "tree_a.urdf"))
    assert [p.kind for p in ps] == ["cylinder"] * 3 and [p.link for p in ps] == ["trunk", "branch_1", "branch_2"]

    def T(xyz, rpy):
        m = np.eye(4)
        m[:3, :3], m[:3, 3] = rpy_matrix(*rpy), xyz
        return m

    j01, j12 = T([0.4, 0.2, 2.2], [0.9, 0.3, -0.5]), T([0.0, 0.3, 0.6], [-0.7, 0.2, 1.1])
    assert np.allclose(ps[0].T, T([0.1, -0.05, 1.5], [0, 0.1, 0.3]))          # root link: only the visual origin
    assert np.allclose(ps[1].T, j01)                                            # child: joint origin
    assert np.allclose(ps[2].T, j01 @ j12 @ T([0, 0, 0.2], [0, 0, 0]))          # grandchild: chain x visual origin
    for p in ps:
        q = quat_xyzw_from_matrix(p.T[:3, :3])
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert np.allclose(R, p.T[:3, :3], atol=1e-12) and w >= 0
        tris = tessellate(p).astype(np.float64)
        # trimesh.creation.cylinder's default: 32 sections, per section (bottom cap, side, side, top cap) = 128 triangles, padded to
        # the scene's chunks of 12 with duplicates of the last one
        assert tris.shape == (132, 3, 3) and np.array_equal(tris[128:], np.repeat(tris[127:128], 4, axis=0))
        tris = tris[:128]
        vol = np.einsum("ij,ij->i", tris[:, 0], np.cross(tris[:, 1], tris[:, 2])).sum() / 6  # closed, outward normals
        r, L = p.dims
        assert abs(vol - 0.5 * 32 * r * r * np.sin(2 * np.pi / 32) * L) < 1e-6 * max(vol, 1)
        # first section, in the order trimesh.creation.revolve emits it (profile (0,-h) (r,-h) (r,h) (0,h), theta 0 -> 2 pi / 32)
        c, s_ = np.cos(2 * np.pi / 32), np.sin(2 * np.pi / 32)
        want = np.array([[[r, 0, -L / 2], [0, 0, -L / 2], [r * c, r * s_, -L / 2]], [[r, 0, -L / 2], [r * c, r * s_, -L / 2], [r, 0, L / 2]],
                         [[r, 0, L / 2], [r * c, r * s_, -L / 2], [r * c, r * s_, L / 2]], [[r, 0, L / 2], [r * c, r * s_, L / 2], [0, 0, L / 2]]])
        assert np.allclose(tris[:4], want, atol=1e-6)
        assert np.abs(tris[..., 2]).max() <= L / 2 + 1e-6 and np.abs(np.linalg.norm(tris[..., :2], axis=-1)).max() <= r + 1e-6
        assert half_extents(p) == (r, r, L / 2)
    mixed = load_urdf_primitives(os.path.join(FIX, "trees", "tree_b.urdf"))
    assert [p.kind for p in mixed] == ["box", "cylinder"] and tessellate(mixed[0]).shape == (12, 3, 3)


def test_multi_primitive_scene_layout():
    """canonical primitive layout, per-env owner indices, padding by duplication, per-link semantic ids"""
    import random

    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config import asset_config as A
    from aerial_gym_simulator_amd.config.env_config import ForestEnvCfg
    from aerial_gym_simulator_amd.env_manager.scene_manager import SceneManager

    class trees(A.tree_asset_params):
        num_assets = 2
        asset_folder = os.path.join(FIX, "trees")

    class Cfg(ForestEnvCfg):
        class env_config:
            include_asset_type = {"trees": True, "objects": True, "bottom_wall": True}
            asset_type_to_dict_map = {"trees": trees, "objects": A.object_asset_params, "bottom_wall": A.bottom_wall}

    random.seed(0)
    sc = SceneManager(Cfg, 6, "cpu", None)
    assert sc.has_prims and sc.num_assets == 38 and sc.keep_in_env_num == 3
    assert sc.num_prims == 1 + 2 * 3 + 35 and sc.num_tris == 12 + 2 * 3 * 132 + 35 * 12
    d = sc._np
    pa = d["prim_asset"]
    assert all(sorted(set(pa[i])) == list(range(38)) for i in range(6))        # every asset owns at least one primitive
    assert (pa[:, 0] == 0).all() and (pa[:, 1:4] == 1).all() and (pa[:, 4:7] == 2).all()  # keep_in_env assets are not shuffled
    assert any(not np.array_equal(pa[0, 7:], pa[i, 7:]) for i in range(1, 6))  # the free objects are, per env
    # a 2-link tree in a 3-primitive slot: the third primitive duplicates the first (same box, same triangles, same id)
    two_link = [(i, s) for i in range(6) for s in (1, 4) if np.array_equal(d["prim_half"][i, s], np.float32([0.25, 0.2, 0.4]))]
    assert two_link, "tree_b.urdf (box stump + pole) was never drawn"
    i, s = two_link[0]
    assert np.array_equal(d["prim_half"][i, s + 2], d["prim_half"][i, s]) and d["prim_sem"][i, s + 2] == d["prim_sem"][i, s]
    t0 = d["tri_local"][i, 12 + (s - 1) * 132: 12 + s * 132]
    assert np.array_equal(t0[:12], t0[12:24]) and np.array_equal(t0[:12], t0[120:132])  # a box in a 132-triangle slot: its 12 triangles repeated
    sem = d["prim_sem"]
    assert sem[0, 0] == 13 and sem[0, 1] == 100 and set(np.diff(sem[0, 1:4])) <= {0, 1}  # floor id, then per-link ids from 100
    # the tree form the scene asks the builder for (include/aerial_gym_hip.h): chunks of 12 triangles + object nodes for the boxes, and
    # -- only where EVERY object is a box -- the tree built over the objects (AGX_BVH_OBJECT_TREE: cylinder chunks traverse slower
    # through its quick subtrees, profiles/r06_scene_refresh_phases.txt)
    from aerial_gym_simulator_amd.config.env_config import EnvWithObstaclesCfg

    assert sc.bvh_prims_per_object == 12 | 0x20000000
    boxes_only = SceneManager(EnvWithObstaclesCfg, 2, "cpu", None)
    assert not boxes_only.has_prims and boxes_only.bvh_prims_per_object == 12 | 0x20000000 | 0x10000000
    assert SceneManager(EnvWithObstaclesCfg, 2, "cpu", None, box_objects=False).bvh_prims_per_object == 12


@pytest.mark.skipif(not os.path.isdir(REF_ASSETS), reason="reference tree not present (GPU box)")
def test_reference_trees_and_thin_sets_load():
    from aerial_gym_simulator_amd.assets import list_urdf_files, load_urdf_primitives

    for sub, n_prims in (("trees", 13), ("thin", 1)):
        files = list_urdf_files(os.path.join(REF_ASSETS, sub))
        for f in files[:25]:
            ps = load_urdf_primitives(os.path.join(REF_ASSETS, sub, f))
            assert len(ps) == n_prims and all(np.isfinite(p.T).all() for p in ps)


def test_empty_scene_has_the_attributes_the_binding_reads():
    """EnvManager._bind reads scene.num_prims / boxes for every env, also the obstacle-free ones"""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.env_config import EmptyEnvCfg
    from aerial_gym_simulator_amd.env_manager.scene_manager import SceneManager

    sc = SceneManager(EmptyEnvCfg, 4, "cpu", None)
    assert (sc.num_assets, sc.num_prims, sc.num_tris, sc.has_prims) == (0, 0, 0, False)


def test_mesh_and_sphere_geometry_ingestion():
    """<mesh> (Wavefront OBJ, binary / ascii STL; stdlib readers) and <sphere> obstacle geometry: the reference's WarpAsset
    takes any trimesh (assets/warp_asset.py:19-136).  The three encodings of the same wedge give the same triangles;
    the primitive frame sits at the centre of the mesh's bounding box (its collision OBB); triangle counts are padded
    to the scene's chunks of 12 with duplicates."""
    from aerial_gym_simulator_amd.assets import half_extents, load_urdf_primitives, num_triangles, tessellate
    from aerial_gym_simulator_amd.assets.urdf_primitives import load_mesh_triangles

    M = os.path.join(FIX, "meshes")
    a, b, c = (load_mesh_triangles(os.path.join(M, f)) for f in ("wedge.obj", "wedge_binary.stl", "wedge_ascii.stl"))
    assert a.shape == (8, 3, 3) and np.allclose(a, b, atol=1e-6) and np.allclose(a, c, atol=1e-6)
    for urdf in ("wedge_obj.urdf", "wedge_stl.urdf"):
        (p,) = load_urdf_primitives(os.path.join(M, urdf))
        assert p.kind == "mesh" and num_triangles(p) == 12
        t = tessellate(p)
        assert t.shape == (12, 3, 3) and np.array_equal(t[8:], np.repeat(t[7:8], 4, axis=0))  # padding = duplicates
        assert np.allclose(half_extents(p), (0.4, 0.32, 0.36))  # scale 0.8 x 0.8 x 1.2 of a 1 x 0.8 x 0.6 wedge
        assert np.allclose(t.reshape(-1, 3).min(0), -np.array(half_extents(p)), atol=1e-6)  # centred on its AABB
        # volume of the prism (signed tetrahedra): outward orientation survived the reader
        vol = sum(np.dot(x[0], np.cross(x[1], x[2])) for x in t[:8].astype(np.float64)) / 6.0
        assert abs(vol - 0.5 * 0.8 * 0.64 * 0.72) < 1e-6
        # primitive frame = visual origin (rotated 0.3 rad about z) shifted to the box centre
        c0 = np.array([0.1, 0, 0.05]) + np.array([[np.cos(0.3), -np.sin(0.3), 0], [np.sin(0.3), np.cos(0.3), 0], [0, 0, 1]]) @ np.array([0.4, 0.32, 0.36])
        assert np.allclose(p.T[:3, 3], c0)
    post, ball = load_urdf_primitives(os.path.join(M, "ball_on_post.urdf"))
    # trimesh.creation.icosphere's default: the icosahedron subdivided 3 times = 1280 triangles (642 distinct vertices), padded to 1284
    assert (post.kind, ball.kind) == ("cylinder", "sphere") and num_triangles(ball) == 1284 and num_triangles(post) == 132
    s = tessellate(ball)
    assert len(np.unique(np.round(s.reshape(-1, 3).astype(np.float64), 5), axis=0)) == 642
    assert np.allclose(np.linalg.norm(s.reshape(-1, 3), axis=1), 0.35, atol=1e-6) and np.allclose(ball.T[:3, 3], (0, 0, 1.0))
    vol = sum(np.dot(x[0], np.cross(x[1], x[2])) for x in s.astype(np.float64)) / 6.0
    assert 0.98 * (4 / 3) * np.pi * 0.35 ** 3 < vol < (4 / 3) * np.pi * 0.35 ** 3  # closed, outward, inscribed
    with pytest.raises(NotImplementedError, match="OBJ, STL"):
        load_mesh_triangles(os.path.join(M, "missing.dae"))


def _forest_scene(tree_folder, n_envs=4):
    import random

    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config import asset_config as A
    from aerial_gym_simulator_amd.config.env_config import ForestEnvCfg
    from aerial_gym_simulator_amd.env_manager.scene_manager import SceneManager

    class trees(A.tree_asset_params):
        asset_folder = tree_folder

    class Cfg(ForestEnvCfg):
        class env_config:
            include_asset_type = {"trees": True, "objects": True, "bottom_wall": True}
            asset_type_to_dict_map = {"trees": trees, "objects": A.object_asset_params, "bottom_wall": A.bottom_wall}

    random.seed(1)
    return SceneManager(Cfg, n_envs, "cpu", None)


def _check_forest_scene_arrays(sc, n_envs):
    """forest_env at the reference's size (forest_env.py + env_object_config.py:225-312): a floor slab, ONE 13-link cylinder tree
    (keep_in_env, per-link semantic ids), 35 box objects.  Scene arrays only -- what assets/warp_asset.py:19-136 and
    warp_env_manager.py:74-95 build: triangle counts per link mesh, the per-link ids 0..12 + the env's running counter."""
    d = sc._np
    assert sc.has_prims and sc.num_assets == 37 and sc.keep_in_env_num == 2
    assert sc.num_prims == 1 + 13 + 35
    # every cylinder = trimesh.creation.cylinder's 128 triangles (+ 4 duplicates filling the last chunk of 12)
    assert sc.num_tris == 12 + 13 * 132 + 35 * 12 == 2148
    tp = d["tri_prim"]
    assert np.array_equal(np.bincount(tp), [12] + [132] * 13 + [12] * 35)
    sem = d["prim_sem"]
    ids_per_env = 13 + 35  # the tree's 13 links + one per object (semantic_id < 0); the floor has its own fixed id
    for i in range(n_envs):
        base = 100 + i * ids_per_env
        assert sem[i, 0] == 13                                             # bottom_wall: configured id
        assert np.array_equal(sem[i, 1:14], base + np.arange(13))          # per_link_semantic: one id per link, in link order
        assert sorted(sem[i, 14:]) == list(range(base + 13, base + 48))    # then the objects, one each (shuffled per env)
    tl = d["tri_local"]
    for q in range(1, 14):  # each tree primitive: all vertices on its cylinder (radius r on the side rings, 0 on the axis)
        t = tl[0, 12 + (q - 1) * 132: 12 + q * 132].reshape(-1, 3).astype(np.float64)
        r, hz = d["prim_half"][0, q, 0], d["prim_half"][0, q, 2]
        rad = np.linalg.norm(t[:, :2], axis=1)
        assert np.all((np.abs(rad - r) < 1e-5) | (rad < 1e-7)) and np.allclose(np.abs(t[:, 2]), hz, atol=1e-5)
        assert len(np.unique(np.round(t, 6), axis=0)) == 2 * 32 + 2


def test_forest_env_scene_arrays_at_reference_size_from_the_fixture_tree():
    """runs everywhere (the GPU box has no reference tree): tests/fixtures/assets/trees13/tree_13.urdf is a synthetic tree with
    the reference set's structure (make_tree13.py)"""
    sc = _forest_scene(os.path.join(FIX, "trees13"))
    _check_forest_scene_arrays(sc, 4)


@pytest.mark.skipif(not os.path.isdir(REF_ASSETS), reason="reference tree not present (GPU box)")
def test_forest_env_scene_arrays_from_the_reference_resources():
    """forest_env built from the reference's own resources/models/environment_assets/trees (100 files, one
    drawn per env) on the CPU box: the same array checks as the fixture tree"""
    sc = _forest_scene(os.path.join(REF_ASSETS, "trees"), n_envs=6)
    _check_forest_scene_arrays(sc, 6)
