from .registries import sim_config_registry  # noqa: F401  (reference module path aerial_gym/registry/sim_registry.py)
