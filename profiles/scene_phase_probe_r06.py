"""Round 6: where the refresh of ONE dirty env spends its time (k_scene_refresh, direct mode, 256 envs, configs[2] scene).
Builds a variant of the library with -DAGX_SCENE_PHASE_CLOCK (thread 0 stamps the 100 MHz wall clock at the phase boundaries),
makes exactly one env dirty per launch and prints the phase durations in microseconds (median of 50 launches).
    python profiles/scene_phase_probe_r06.py"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "aerial_gym_simulator_amd", "lib", "libagx_phaseclock.so")

if os.environ.get("AGX_LIB_PATH") != VARIANT:
    from aerial_gym_simulator_amd import _build

    _build.build_library(extra_flags=["-DAGX_SCENE_PHASE_CLOCK"], lib_path=VARIANT)
    raise SystemExit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, AGX_LIB_PATH=VARIANT)))

import torch  # noqa: E402

import bench  # noqa: E402
from aerial_gym_simulator_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
task = bench.make_task("depth", n, "cuda:0", False)
task.reset()
env = task.sim_env
lib = _lib.load()
raw = C.CDLL(VARIANT)
g = env.global_tensor_dict
a = torch.zeros(n, 4, device="cuda:0")
for _ in range(3):
    task.step(a)
torch.cuda.synchronize()
names = ["asset reset", "transform", "boxes + barrier", "phase 1 (bounds, box verdicts) + barrier", "phase 2a (object keys)", "phase 2b (rank sort)",
         "phase 2c (radix tree + emit)"]
rows, waves, split = [], [], []
start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
kernel_us = []
for rep in range(50):
    g["reset_mask"].zero_()
    g["reset_mask"][rep % n] = 1
    g["reset_flag"][env._parity] = 1
    torch.cuda.synchronize()
    start.record()
    env.asset_manager.reset_masked(env)
    stop.record()
    torch.cuda.synchronize()
    kernel_us.append(start.elapsed_time(stop) * 1e3)
    out = (C.c_ulonglong * 16)()
    assert raw.agx_debug_phase_clock(out) == 0
    t = [out[k] for k in range(11)]
    rows.append([(t[k + 1] - t[k]) / 100.0 for k in range(7)])
    split.append(((t[9] - t[6]) / 100.0, (t[10] - t[9]) / 100.0, (t[7] - t[10]) / 100.0))
    waves.append((t[8] - t[4]) / 100.0)
med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
res = {"num_envs": n, "launch_us_events": med(kernel_us), "phases_us": {nm: med([r[k] for r in rows]) for k, nm in enumerate(names)},
       "phase 3 (object records, waves 1-7, from the barrier)": med(waves),
       "phase 2c split (tree, box propagation, emit)": [med([x[k] for x in split]) for k in range(3)], "sum_us": sum(med([r[k] for r in rows]) for k in range(7))}
print(json.dumps(res))
