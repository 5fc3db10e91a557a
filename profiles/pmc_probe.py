"""Workload for the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE): a handful of launches of
  - k_update_states at 2^21 envs   (calibration: known 52 B read + 64 B written per env,
                                     same 4 B/lane coalesced SoA pattern as the step kernel)
  - k_env_step<4,position> at 8192 and 2^21 envs
  - (with --nav) three navigation steps at 8192 envs for k_raycast / k_env_step<4,velocity,10 sub-steps>
Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python profiles/pmc_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = "cuda:0"
for n in (8192, 1 << 21):
    task = bench.make_task("dynamics", n, dev, False, lean=False)  # every dict tensor maintained (the key of this grid size)
    task.reset()
    a = torch.rand(n, 4, device=dev) * 2 - 1
    for _ in range(5):
        task.step(a)
    env = task.sim_env
    torch.cuda.synchronize()
    for _ in range(5):
        env.update_states()
    torch.cuda.synchronize()
    del task
# the lean launch of the same kernel (its own env count: the counters are keyed by kernel name + grid size)
task = bench.make_task("dynamics", bench.LEAN_AT_SCALE_ENVS, dev, False, lean=True)
task.reset()
a = torch.rand(bench.LEAN_AT_SCALE_ENVS, 4, device=dev) * 2 - 1
for _ in range(5):
    task.step(a)
torch.cuda.synchronize()
del task
if "--nav" in sys.argv:
    t = bench.make_task("depth", 8192, dev, False)
    t.reset()
    a = torch.rand(8192, 4, device=dev) * 2 - 1
    for _ in range(3):
        t.step(a)
    torch.cuda.synchronize()
    del t
    t = bench.make_task("lidar", 4096, dev, False)  # BASELINE configs[3]
    t.reset()
    a = torch.rand(4096, 4, device=dev) * 2 - 1
    for _ in range(3):
        t.step(a)
    torch.cuda.synchronize()
print("pmc probe done")
from aerial_gym_simulator_amd import _lib  # noqa: E402

with open(os.environ.get("AGX_BUILD_ID_OUT", os.path.join(ROOT, "gpurun_out", "pmc_build_id.txt")), "w") as f:
    f.write(_lib.build_id())  # which binary the counters belong to (collect_pmc.py stamps pmc_traffic.json with it)
