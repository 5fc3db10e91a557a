#!/bin/bash
# Round 4 EXPERIMENT (not in the product: apply profiles/src/lidar_nav_raycast_epilogue_experiment_r04.patch first; results in
# profiles/r04_lidar_nav_epilogue_experiment.txt): the LiDAR-navigation epilogue of the ray-cast kernel (agx_raycast_lidar_nav +
# agx_lidar_image_obs_from_range) -- its tests, and the task's step with and without it (bench line + rocprofv3 kernel statistics).
#   gpurun --timeout 900 -- 'bash profiles/measure_r04_m.sh'
set -u
O=gpurun_out/r04n
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lidar_nav.py tests/test_gpu_raycast.py -m gpu -q -x > $O/pytest_lidar_nav.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_lidar_nav.log
for ep in 1 0; do
  AGX_LIDAR_NAV_EPILOGUE=$ep python bench.py --workload lidar_nav --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_lidar_nav_ep$ep.json 2>/dev/null
  AGX_LIDAR_NAV_EPILOGUE=$ep rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lidar_nav_ep$ep -o p -- python bench.py --workload lidar_nav --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/prof_lidar_nav_ep$ep.err
done
python - <<'P'
import json, csv
for ep in (1, 0):
    d = json.loads(open("gpurun_out/r04n/bench_lidar_nav_ep%d.json" % ep).read().strip().splitlines()[-1])
    print("epilogue", ep, "lidar_nav", d["value"], d["ms_per_step"])
    for r in csv.DictReader(open("gpurun_out/r04n/prof_lidar_nav_ep%d/p_kernel_stats.csv" % ep)):
        if "k_raycast" in r["Name"] or "k_lidar_image" in r["Name"]:
            print("  %-50s %8.1f us %s%%" % (r["Name"].split("(")[0][-50:], float(r["AverageNs"]) / 1e3, r["Percentage"]))
P
