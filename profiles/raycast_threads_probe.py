"""k_raycast: workgroup size (AGX_RAY_THREADS = 64 / 128 / 256 / 512) under the XCD-aware mapping, each variant in its own process
with the launch policy's split (AGX_RAY_SPLIT unset).   python profiles/raycast_threads_probe.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--child" in sys.argv:
    import torch

    import bench

    which = sys.argv[sys.argv.index("--child") + 1]
    n = 8192 if which == "depth" else 4096
    task = bench.make_task(which, n, "cuda:0", False)
    task.reset()
    a = torch.rand(n, task.task_config.action_space_dim, device="cuda:0") * 2 - 1
    for _ in range(3):
        task.step(a)
    torch.cuda.synchronize()
    ts = sorted(bench.kernel_time_raycast(task, reps=10) * 1e6 for _ in range(5))
    print(json.dumps({"variant": os.environ.get("AGX_VARIANT"), "workload": which, "raycast_us_median": ts[2], "min": ts[0], "max": ts[-1]}), flush=True)
    sys.exit(0)

from aerial_gym_simulator_amd import _build  # noqa: E402

for threads in (256, 64, 128, 512):
    env = dict(os.environ, AGX_VARIANT="threads%d" % threads)
    if threads != 256:
        lib = os.path.join(ROOT, "aerial_gym_simulator_amd", "lib", "libaerialgym_hip_ray%d.so" % threads)
        if not os.path.exists(lib):
            _build.build_library(extra_flags=["-DAGX_RAY_THREADS=%d" % threads], lib_path=lib)
        env["AGX_LIB_PATH"] = lib
    for which in ("depth", "lidar"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", which], env=env, capture_output=True, text=True, timeout=600)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-400:], flush=True)
