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
