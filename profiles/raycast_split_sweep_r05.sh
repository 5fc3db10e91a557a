mkdir -p gpurun_out/r05s
for s in 1 2 3 4 6 12; do SPLIT=$s python - <<PY >> gpurun_out/r05s/sweep.txt 2>/dev/null
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from aerial_gym_simulator_amd import _lib; _lib.set_option("ray_split", int(os.environ["SPLIT"]))
n=8192
task = bench.make_task("depth", n, "cuda:0", False); task.reset()
g = torch.Generator(device="cuda:0").manual_seed(4321)
acts = [torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1 for _ in range(4)]
for i in range(12): task.step(acts[i % 4])
torch.cuda.synchronize()
print("camera split", os.environ["SPLIT"], "us %.1f" % (min(bench.kernel_time_raycast(task) for _ in range(2)) * 1e6))
PY
done
for s in 8 16 32 64 128; do SPLIT=$s python - <<PY >> gpurun_out/r05s/sweep.txt 2>/dev/null
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from aerial_gym_simulator_amd import _lib; _lib.set_option("ray_split", int(os.environ["SPLIT"]))
n=4096
task = bench.make_task("lidar", n, "cuda:0", False); task.reset()
g = torch.Generator(device="cuda:0").manual_seed(4321)
acts = [torch.rand(n, 7, device="cuda:0", generator=g) * 2 - 1 for _ in range(4)]
for i in range(12): task.step(acts[i % 4])
torch.cuda.synchronize()
print("lidar split", os.environ["SPLIT"], "us %.1f" % (min(bench.kernel_time_raycast(task) for _ in range(2)) * 1e6))
PY
done
cat gpurun_out/r05s/sweep.txt
