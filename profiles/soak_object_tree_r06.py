"""Round 6 soak: the object-level BVH build against the triangle-level LBVH (no object nodes) over many random scenes.
Two navigation tasks with the same seeds step side by side; every frame (depth, segmentation), every observation, reward and flag
must be bit-identical -- the tree is a pure accelerator.  ~5 % of the envs reset per step under random actions: N x steps x 0.05
scene rebuilds.     python profiles/soak_object_tree_r06.py [num_envs] [steps] [workload]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
workload = sys.argv[3] if len(sys.argv) > 3 else "depth"
dev = "cuda:0"
tasks = []
for box in (True, False):
    t = bench.make_task(workload, n, dev, False, obstacles="curriculum" if len(sys.argv) > 4 else "all", extra_args={"bvh_box_objects": box, "rng_seed": 1234})
    t.reset()
    tasks.append(t)
a_obj, a_tri = tasks
A = a_obj.task_config.action_space_dim
g = torch.Generator(device=dev).manual_seed(7)
bad = torch.zeros((), dtype=torch.int64, device=dev)
t0 = time.perf_counter()
first_bad = None
for s in range(steps):
    a = torch.rand(n, A, device=dev, generator=g) * 2 - 1
    o1 = a_obj.step(a)
    o2 = a_tri.step(a)
    same = (torch.equal(a_obj.obs_dict["depth_range_pixels"].view(torch.int32), a_tri.obs_dict["depth_range_pixels"].view(torch.int32))
            and torch.equal(a_obj.obs_dict["segmentation_pixels"], a_tri.obs_dict["segmentation_pixels"])
            and torch.equal(o1[0]["observations"].view(torch.int32), o2[0]["observations"].view(torch.int32))
            and torch.equal(o1[1].view(torch.int32), o2[1].view(torch.int32)) and torch.equal(o1[2], o2[2]) and torch.equal(o1[3], o2[3]))
    if not same and first_bad is None:
        first_bad = s
        break
resets = int(a_obj.sim_env.global_tensor_dict["episode_count"].sum()) - n
print(json.dumps({"workload": workload, "num_envs": n, "steps": s + 1, "scene_rebuilds": resets, "bit_identical": first_bad is None,
                  "first_difference_at_step": first_bad, "seconds": round(time.perf_counter() - t0, 1)}))
