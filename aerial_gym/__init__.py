"""`aerial_gym` -- import-path alias of `aerial_gym_simulator_amd`, so that code written against the reference
(ntnu-arl/aerial_gym_simulator: its rl_training scripts, examples, user task files) imports unchanged:

    from aerial_gym.registry.task_registry import task_registry      # reference rl_games/runner.py:9
    from aerial_gym.utils.helpers import parse_arguments              # reference rl_games/runner.py:10
    from aerial_gym.sim.sim_builder import SimBuilder                 # reference examples/*.py

Every `aerial_gym.x.y` resolves to THE SAME module object as `aerial_gym_simulator_amd.x.y` (one set of registries,
one library handle), through a meta-path alias finder; nothing is imported twice.  The package directory must be on
sys.path (repo root, or `pip install -e .`).  Together with the inert `isaacgym` package next to it this is what
"drop-in for the rl_training scripts" means; INTEGRATION.md lists what is and is not covered."""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys

import aerial_gym_simulator_amd as _impl

_ALIAS, _REAL = __name__, _impl.__name__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_ALIAS + "."):
            return None
        real = _REAL + fullname[len(_ALIAS):]
        try:
            mod = importlib.import_module(real)
        except ModuleNotFoundError as e:
            if e.name == real:  # the reference has sub-packages this path does not rebuild
                return None
            raise
        spec = importlib.machinery.ModuleSpec(fullname, self, is_package=hasattr(mod, "__path__"))
        spec._agx_module = mod
        return spec

    def create_module(self, spec):
        return spec._agx_module  # the implementation module itself: aliases share state

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

AERIAL_GYM_DIRECTORY = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # aerial_gym/__init__.py:4
for _name in ("registry", "config", "control", "robots", "env_manager", "sensors", "sim", "task", "utils", "assets"):
    globals()[_name] = importlib.import_module(_ALIAS + "." + _name)
