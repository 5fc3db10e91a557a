import sys, json, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'.')
import bench
wl, n = sys.argv[1], int(sys.argv[2])
task=bench.make_task(wl, n, "cuda:0", False, 0)
task.reset()
A=task.task_config.action_space_dim
g=torch.Generator(device="cuda:0").manual_seed(1)
acts=[torch.rand(n,A,device="cuda:0",generator=g)*2-1 for _ in range(8)]
for i in range(30): task.step(acts[i%8])
torch.cuda.synchronize()
t0=time.time()
for i in range(300): task.step(acts[i%8])
torch.cuda.synchronize()
print(json.dumps({"workload": wl, "num_envs": n, "us_per_step": (time.time()-t0)/300*1e6}))
