"""Inert stand-in for NVIDIA Isaac Gym, for code that does `import isaacgym` / `from isaacgym import gymutil` before
importing `aerial_gym` (reference rl_games/runner.py:6, cleanrl/ppo_continuous_action.py:36-37, utils/helpers.py:31-32).

aerial_gym_simulator_amd never calls Isaac Gym: rigid-body integration, contacts and ray-casting run in its own HIP
kernels.  This package only provides the handful of NAMES the reference's trainers and helpers touch at import / argument
parsing time (gymutil.parse_device_str, gymapi.SIM_PHYSX, gymapi.SimParams, ...).  Anything that would need the real
simulator raises.  If the real Isaac Gym is installed and comes first on sys.path it is used instead (and ignored by
this framework all the same)."""
from . import gymapi, gymtorch, gymutil  # noqa: F401

__all__ = ["gymapi", "gymtorch", "gymutil"]
INERT_STUB = True
