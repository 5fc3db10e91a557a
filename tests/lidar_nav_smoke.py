import sys, time; sys.path.insert(0, "/root/repo")
import torch
import aerial_gym_simulator_amd  # noqa
from aerial_gym_simulator_amd.config.task_config import lidar_navigation_task_config as cfg
from aerial_gym_simulator_amd.registry.task_registry import task_registry
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg.device = "cuda:0"
task = task_registry.make_task("lidar_navigation_task", seed=1, num_envs=n, headless=True)
obs, r, term, trunc, info = task.reset()
print("obs", obs["observations"].shape, "finite", bool(torch.isfinite(obs["observations"]).all()))
a = torch.rand(n, 4, device="cuda:0") * 2 - 1
for i in range(30):
    obs, r, term, trunc, info = task.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 100
nc = nt = 0
for i in range(K):
    obs, r, term, trunc, info = task.step(a)
    nc += int(term.sum()); nt += int(trunc.sum())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
o = obs["observations"]
print("finite", bool(torch.isfinite(o).all()), "reward mean", float(r.mean()), "crashes", nc, "truncs", nt)
print("ttc min/mean", float(task.time_to_collision.min()), float(task.time_to_collision.mean()), "inv range min/max", float(o[:, 17:].min()), float(o[:, 17:].max()))
print("env-steps/s", n * K / dt, "ms/step", 1e3 * dt / K)
