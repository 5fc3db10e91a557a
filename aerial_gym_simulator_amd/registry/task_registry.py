from .registries import task_registry  # noqa: F401  (reference module path aerial_gym/registry/task_registry.py)
