from .registries import controller_registry  # noqa: F401  (reference module path aerial_gym/registry/controller_registry.py)
