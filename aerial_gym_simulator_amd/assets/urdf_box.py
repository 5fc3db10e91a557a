"""Box obstacles from URDF files (the reference's WarpAsset / asset_loader for its shipped obstacle sets).

Every obstacle the reference ships for `env_with_obstacles` / `env_with_lidar_nav_obstacles` / `dynamic_env`
(resources/models/environment_assets/{panels,objects,walls}/*.urdf) is a single link with one <box>.  This
module reads such files (stdlib XML, no urdfpy / trimesh) and is what tests/test_assets.py uses to check the
box-size tables restated in config/asset_config.py against the reference's files.  The scene builder itself goes
through assets/urdf_primitives.py, which also handles multi-link box / cylinder assets (`trees`, `thin`);
parse_box_urdf stays strict: anything but one box in one link raises NotImplementedError."""
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass


@dataclass(frozen=True)
class BoxAsset:
    file: str
    size: tuple        # (x, y, z) extents of the box [m]
    origin_xyz: tuple  # visual origin in the link frame
    origin_rpy: tuple


def _floats(text, n, what, path):
    vals = [float(v) for v in text.split()]
    if len(vals) != n:
        raise ValueError(f"{path}: {what} needs {n} numbers, got '{text}'")
    return tuple(vals)


def parse_box_urdf(path, use_collision=False):
    """The single <box> of a single-link URDF.  use_collision: read <collision> instead of <visual>
    (asset option use_collision_mesh_instead_of_visual, warp_asset.py:33-45)."""
    root = ET.parse(path).getroot()
    links = root.findall("link")
    if len(links) != 1:
        raise NotImplementedError(f"{path}: {len(links)} links; only single-link box obstacles are supported (SURVEY 8 f3)")
    elems = links[0].findall("collision" if use_collision else "visual")
    if len(elems) != 1:
        raise NotImplementedError(f"{path}: expected exactly one {'collision' if use_collision else 'visual'} element")
    geom = elems[0].find("geometry")
    box = geom.find("box") if geom is not None else None
    if box is None:
        kinds = [c.tag for c in geom] if geom is not None else []
        raise NotImplementedError(f"{path}: geometry {kinds} is not a box; only box obstacles are supported (SURVEY 8 f3)")
    size = _floats(box.get("size"), 3, "box size", path)
    if min(size) <= 0.0:
        raise ValueError(f"{path}: non-positive box size {size}")
    origin = elems[0].find("origin")
    xyz = _floats(origin.get("xyz", "0 0 0"), 3, "origin xyz", path) if origin is not None else (0.0, 0.0, 0.0)
    rpy = _floats(origin.get("rpy", "0 0 0"), 3, "origin rpy", path) if origin is not None else (0.0, 0.0, 0.0)
    if any(abs(v) > 1e-12 for v in xyz + rpy):
        raise NotImplementedError(f"{path}: box with a non-identity visual origin {xyz} {rpy} is not supported yet")
    return BoxAsset(os.path.basename(path), size, xyz, rpy)


def list_urdf_files(folder):
    """asset_loader.py:44-53: the *.urdf files directly inside `folder` (os.listdir order)."""
    return [f for f in os.listdir(folder) if f.endswith(".urdf")]
