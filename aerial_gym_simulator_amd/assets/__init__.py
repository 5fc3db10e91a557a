"""Obstacle asset ingestion (aerial_gym/assets, env_manager/asset_loader.py)."""
from .urdf_box import BoxAsset, list_urdf_files, parse_box_urdf  # noqa: F401
