"""Obstacle (asset) descriptions.

Every obstacle the reference ships for `env_with_obstacles` is a single-link URDF box
(resources/models/environment_assets/{panels,objects,walls}/*.urdf), so an asset type is
described here by its box sizes + the placement ranges of
aerial_gym/config/asset_config/env_object_config.py.  URDF/mesh ingestion is out of scope
(SURVEY.md section 8 f3).
"""
import os

import numpy as np

_PI = float(np.pi)
# Root of the reference's (or your own) `resources` tree for asset types that are read from URDF files.  The
# box-only sets are restated below as data and need no files; `trees` / `thin` do.
RESOURCES = os.environ.get("AERIAL_GYM_RESOURCES", "/nonexistent/aerial_gym/resources")


def _assets(sub):
    return os.path.join(RESOURCES, "models", "environment_assets", sub)

PANEL_SEMANTIC_ID = 20
FRONT_WALL_SEMANTIC_ID, BACK_WALL_SEMANTIC_ID = 9, 10
LEFT_WALL_SEMANTIC_ID, RIGHT_WALL_SEMANTIC_ID = 11, 12
BOTTOM_WALL_SEMANTIC_ID, TOP_WALL_SEMANTIC_ID = 13, 14


def _ratio(lo_xyz, hi_xyz, lo_rpy=(0.0, 0.0, 0.0), hi_rpy=(0.0, 0.0, 0.0)):
    tail = [1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    return list(lo_xyz) + list(lo_rpy) + tail, list(hi_xyz) + list(hi_rpy) + tail


class asset_state_params:
    num_assets = 1
    # Geometry source, in this order:
    #  1. asset_folder (+ file): URDF files as in the reference's config (env_object_config.py:20-21): `file` for
    #     every instance, or file = None -> random.choices over the folder's *.urdf (asset_loader.py:44-56).  Used
    #     when the folder exists on this machine (e.g. <aerial_gym>/resources/models/environment_assets/objects).
    #  2. box_sizes: the same boxes restated as data (checked against the reference's URDFs by
    #     tests/test_assets.py when the reference tree is present); one entry is picked per instance.
    asset_folder = None
    file = None
    use_collision_mesh_instead_of_visual = False
    box_sizes = [[1.0, 1.0, 1.0]]
    random_box_size_range = None   # or ([lo xyz], [hi xyz]): sizes drawn per instance
    keep_in_env = False
    semantic_id = -1               # < 0: assigned incrementally per instance (from 100)
    min_state_ratio, max_state_ratio = _ratio([0.5] * 3, [0.5] * 3)
    color = None


class panel_asset_params(asset_state_params):  # env_object_config.py:65-120, panels/panel.urdf
    num_assets = 3
    box_sizes = [[0.1, 1.2, 3.0]]
    keep_in_env = True
    min_state_ratio, max_state_ratio = _ratio([0.3, 0.05, 0.05], [0.85, 0.95, 0.95], (0, 0, -_PI / 3), (0, 0, _PI / 3))
    color = [170, 66, 66]


class object_asset_params(asset_state_params):  # env_object_config.py:273-312, objects/*.urdf
    num_assets = 35
    box_sizes = [[0.1, 0.5, 0.5], [0.1, 1.0, 1.0], [0.1, 0.1, 2.0], [0.4, 0.4, 0.4]]
    keep_in_env = False
    min_state_ratio, max_state_ratio = _ratio([0.30, 0.05, 0.05], [0.85, 0.9, 0.9], (-_PI,) * 3, (_PI,) * 3)


class random_box_asset_params(asset_state_params):
    """Synthetic obstacle set of BASELINE config 3/4: 100 boxes,
    every side U(0.1, 1.2) m, centre anywhere in the bounds, yaw U(-pi, pi)."""

    num_assets = 100
    box_sizes = None
    random_box_size_range = ([0.1, 0.1, 0.1], [1.2, 1.2, 1.2])
    keep_in_env = False
    min_state_ratio, max_state_ratio = _ratio([0.0, 0.0, 0.0], [1.0, 1.0, 1.0], (0, 0, -_PI), (0, 0, _PI))


def _wall(name, size, ratio_xyz, sem_id):
    lo, hi = _ratio(ratio_xyz, ratio_xyz)
    return type(name, (asset_state_params,), dict(num_assets=1, box_sizes=[size], keep_in_env=True, semantic_id=sem_id,
                                                  min_state_ratio=lo, max_state_ratio=hi, color=[100, 200, 210]))


# env_object_config.py:316-598 and walls/*.urdf
left_wall = _wall("left_wall", [20.0, 0.2, 20.0], [0.5, 1.0, 0.5], LEFT_WALL_SEMANTIC_ID)
right_wall = _wall("right_wall", [20.0, 0.2, 20.0], [0.5, 0.0, 0.5], RIGHT_WALL_SEMANTIC_ID)
top_wall = _wall("top_wall", [20.0, 20.0, 0.2], [0.5, 0.5, 1.0], TOP_WALL_SEMANTIC_ID)
bottom_wall = _wall("bottom_wall", [20.0, 20.0, 0.2], [0.5, 0.5, 0.0], BOTTOM_WALL_SEMANTIC_ID)
front_wall = _wall("front_wall", [0.2, 20.0, 20.0], [1.0, 0.5, 0.5], FRONT_WALL_SEMANTIC_ID)
back_wall = _wall("back_wall", [0.2, 20.0, 20.0], [0.0, 0.5, 0.5], BACK_WALL_SEMANTIC_ID)


# ---- env_with_lidar_nav_obstacles (config/asset_config/lidar_nav_env_config.py): the same URDF boxes, more of
# them, spread over the whole (larger) env, and NOTHING is kept in the env unconditionally -- walls included:
# the curriculum level decides how many of the shuffled assets stay (the rest is parked at -1000 m).
class lidar_nav_panel_asset_params(panel_asset_params):  # lidar_nav_env_config.py:65-120
    num_assets = 15
    keep_in_env = False
    min_state_ratio, max_state_ratio = _ratio([0.35, 0.0, 0.0], [1.0, 1.0, 1.0], (0, 0, -_PI / 3), (0, 0, _PI / 3))


class lidar_nav_object_asset_params(object_asset_params):  # lidar_nav_env_config.py:273-312
    num_assets = 70
    keep_in_env = False
    min_state_ratio, max_state_ratio = _ratio([0.30, 0.0, 0.0], [1.0, 1.0, 1.0], (-_PI,) * 3, (_PI,) * 3)


def _free(wall):
    return type("lidar_nav_" + wall.__name__, (wall,), dict(keep_in_env=False))  # lidar_nav_env_config.py:355-592


lidar_nav_walls = {name: _free(w) for name, w in (("left_wall", left_wall), ("right_wall", right_wall), ("back_wall", back_wall),
                                                  ("front_wall", front_wall), ("bottom_wall", bottom_wall), ("top_wall", top_wall))}


# ---- sets that only exist as URDF files (multi-link cylinder trees, thin rods): AERIAL_GYM_RESOURCES must point
# at a `resources` directory that has models/environment_assets/{trees,thin}
class tree_asset_params(asset_state_params):  # env_object_config.py:225-270
    num_assets = 1
    asset_folder = _assets("trees")
    box_sizes = None
    keep_in_env = True
    per_link_semantic = True
    semantic_id = -1
    min_state_ratio, max_state_ratio = _ratio([0.1, 0.1, 0.0], [0.9, 0.9, 0.0], (0, -_PI / 6, -_PI), (0, _PI / 6, _PI))
    color = [70, 200, 100]


class thin_asset_params(asset_state_params):  # env_object_config.py:181-222
    num_assets = 0
    asset_folder = _assets("thin")
    box_sizes = None
    min_state_ratio, max_state_ratio = _ratio([0.3, 0.05, 0.05], [0.85, 0.95, 0.95], (-_PI,) * 3, (_PI,) * 3)
    color = [170, 66, 66]
