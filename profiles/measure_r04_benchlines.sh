#!/bin/bash
# Round 4: the bench lines again, on the SAME binary as profiles/measure_r04.sh, now that profiles/pmc_traffic.json and
# profiles/r04_valu_peak.json of that pass are in the tree (bench.py reads its traffic / instruction counts / class mix from them;
# the lines of the pass itself were written against the previous build's file and say so: traffic_stale).
#   gpurun --timeout 900 -- 'bash profiles/measure_r04_benchlines.sh'   then   bash profiles/collect_r04.sh
set -u
O=gpurun_out/r04m
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python bench.py --workload depth --steps 200 --warmup 20 > $O/bench_depth.json 2>/dev/null
python bench.py --workload lidar --steps 100 --warmup 10 > $O/bench_lidar.json 2>/dev/null
python bench.py --workload lidar_velocity --steps 100 --warmup 10 > $O/bench_lidar_velocity.json 2>/dev/null
python bench.py --workload lidar_nav --steps 200 --warmup 20 > $O/bench_lidar_nav.json 2>/dev/null
AGX_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline > $O/bench_forced_dist_world1.json 2> $O/bench_forced_dist.err
python - <<'P'
import json
for f in ("bench_default", "bench_driver_style", "bench_depth", "bench_lidar", "bench_lidar_velocity", "bench_lidar_nav", "bench_forced_dist_world1"):
    try:
        d = json.loads(open("gpurun_out/r04m/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"].get("traffic_stale"), d["roofline"].get("counters_stale"))
    except Exception as e:
        print(f, "unreadable", e)
P
