"""Obstacle asset ingestion (aerial_gym/assets, env_manager/asset_loader.py)."""
from .urdf_box import BoxAsset, list_urdf_files, parse_box_urdf  # noqa: F401
from .urdf_primitives import Prim, half_extents, load_urdf_primitives, num_triangles, quat_xyzw_from_matrix, tessellate  # noqa: F401,E402
