import sys, time, torch
sys.path.insert(0, '.')
import aerial_gym_simulator_amd
from aerial_gym_simulator_amd.config.task_config import navigation_task_config as c
from aerial_gym_simulator_amd.registry.task_registry import task_registry
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
c.device = 'cuda:0'; c.args = {"strict_rng": True}
t0 = time.time()
t = task_registry.make_task('navigation_task', seed=1, num_envs=n)
print("build", time.time() - t0)
t.reset()
a = torch.rand(n, 4, device='cuda:0') * 2 - 1
for i in range(20):
    obs, rew, term, trunc, info = t.step(a)
torch.cuda.synchronize()
t0 = time.time(); K = 50
for i in range(K):
    obs, rew, term, trunc, info = t.step(a)
torch.cuda.synchronize()
dt = time.time() - t0
px = t.obs_dict["depth_range_pixels"]; seg = t.obs_dict["segmentation_pixels"]
print("env-steps/s", n * K / dt, "ms/step", 1e3 * dt / K)
print("depth min/max/mean", float(px.min()), float(px.max()), float(px.mean()), "hit frac", float((seg != -2).float().mean()))
print("rew mean", float(rew.mean()), "crash frac", float(term.float().mean()), "obs finite", bool(torch.isfinite(obs["observations"]).all()))
