from .registries import env_config_registry  # noqa: F401  (reference module path aerial_gym/registry/env_registry.py)
