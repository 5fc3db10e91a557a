"""Times the frame's ray-cast launch of BASELINE configs[2] / configs[3] for a list of library builds (AGX_LIB_PATH), with and
without object nodes (args={"bvh_box_objects": ...}): `python profiles/raycast_variant_time.py lib1.so lib2.so ...` -> one line per (library, workload,
box objects): launch us (best of 3 x 20 launches in the step's own state).  Each measurement is its own process."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
import bench
w = sys.argv[1]
n = 4096 if w == "lidar" else 8192
task = bench.make_task(w, n, "cuda:0", False, extra_args={"bvh_box_objects": sys.argv[2] == "1"})
task.reset()
A = task.task_config.action_space_dim
g = torch.Generator(device="cuda:0").manual_seed(4321)
acts = [torch.rand(n, A, device="cuda:0", generator=g) * 2 - 1 for _ in range(4)]
for i in range(12):
    task.step(acts[i %% 4])
torch.cuda.synchronize()
kt = min(bench.kernel_time_raycast(task) for _ in range(2))
dt = bench.timed_steps(task, acts, 40, 5, 1)
print(json.dumps({"raycast_us": kt * 1e6, "ms_per_step": 1e3 * dt / 40}))
''' % ROOT

libs = sys.argv[1:] or [""]
for lib in libs:
    lib, _, modes = lib.partition(":")  # "path.so:0" = triangle subtrees only (a build that does not know object nodes)
    for w in ("depth", "lidar"):
        for box in (modes or "10"):
            env = dict(os.environ)
            if lib:
                env["AGX_LIB_PATH"] = os.path.abspath(lib)
            r = subprocess.run([sys.executable, "-c", CHILD, w, box], env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
            print(os.path.basename(lib) or "default", w, "box_objects", box, json.dumps(out), flush=True)
