"""Step time of the navigation-type tasks at RL-sized batches (256 .. 1024 envs), eager vs replayed hipGraph.
    python profiles/small_batch_r02.py            (GPU box) -> gpurun_out/r02_small_batch.txt"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "one":
    import torch

    import bench

    wl, n = sys.argv[2], int(sys.argv[3])
    task = bench.make_task(wl, n, "cuda:0", False, obstacles="curriculum", extra_args={"step_graph": sys.argv[4] == "1"})
    task.reset()
    A = task.task_config.action_space_dim
    acts = [torch.rand(n, A, device="cuda:0") * 2 - 1 for _ in range(8)]
    for i in range(40):
        task.step(acts[i % 8])
    torch.cuda.synchronize()
    steps = 1500
    t0 = time.perf_counter()
    for i in range(steps):
        task.step(acts[i % 8])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": wl, "num_envs": n, "graph": bool(task._graphs), "us_per_step": 1e6 * dt / steps, "env_steps_per_s": n * steps / dt,
                      "graphs_captured": len(task._graphs) if task._graphs else 0}))
else:
    rows = []
    for wl in ("depth", "lidar_nav"):
        for n in (256, 512, 1024):
            for graph in ("0", "1"):
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "one", wl, str(n), graph], capture_output=True, text=True)
                line = [l for l in out.stdout.splitlines() if l.startswith("{")]
                rows.append(line[-1] if line else json.dumps({"workload": wl, "num_envs": n, "graph": graph, "error": out.stderr[-400:]}))
                print(rows[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "r02_small_batch.txt"), "w").write("\n".join(rows) + "\n")
