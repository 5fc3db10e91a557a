from .registries import robot_registry  # noqa: F401  (reference module path aerial_gym/registry/robot_registry.py)
