"""Round 6 soak: the strict-RNG fast path (agx_position_task_step_strict: mapped host word + one launch reproducing torch's
uniform_ calls) against the general strict path (dispatcher calls, .item()) over many steps with resets in every step: a checksum of
the observation / reward bits per step and the generator offset after every step must agree.
    python profiles/soak_strict_r06.py [num_envs] [steps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dev = "cuda:0"
runs = []
for general in (False, True):
    torch.manual_seed(99)
    t = bench.make_task("dynamics", n, dev, True)
    assert t._strict is not None
    if general:
        t._plan = None
    t.reset()
    bench.desynchronise_episodes(t)
    g = torch.Generator(device=dev).manual_seed(3)
    gen = torch.cuda.default_generators[0]
    sums = torch.zeros(steps, dtype=torch.int64, device=dev)
    offs = []
    for s in range(steps):
        a = torch.rand(n, 4, device=dev, generator=g) * 2 - 1
        obs, rew, term, trunc, _ = t.step(a)
        sums[s] = obs["observations"].view(torch.int32).long().sum() + rew.view(torch.int32).long().sum() + term.long().sum() * 7 + trunc.long().sum() * 13
        offs.append(gen.get_offset())
    runs.append((sums.cpu(), offs, int(t.sim_env.global_tensor_dict["episode_count"].sum()) - n))
    del t
same = bool(torch.equal(runs[0][0], runs[1][0])) and runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2]
print(json.dumps({"num_envs": n, "steps": steps, "resets": runs[0][2], "steps_with_draws": len(set(runs[0][1])), "bit_identical": same}))
