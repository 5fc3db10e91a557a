"""GPU: the runnable counterparts of the reference's example scripts (examples/*.py) run to completion -- the user-facing
loops (raw EnvManager benchmark with and without rendering, position control, RL env loop, navigation task, dynamic
obstacles, IMU logging) stay in working order as the internals change."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args", [
    ("benchmark.py", ["--steps", "200"]),
    ("benchmark.py", ["--rendering", "--steps", "50"]),
    ("position_control_example.py", []),
    ("rl_env_example.py", []),
    ("navigation_task_example.py", []),
    ("dynamic_env_example.py", []),
    ("imu_data_collection.py", []),
])
def test_example_runs(script, args):
    env = dict(os.environ, AGX_EXAMPLE_STEPS="120", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)] + args, env=env, capture_output=True, text=True, timeout=300,
                         cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
