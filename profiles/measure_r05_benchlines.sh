#!/bin/bash
# Round 5, pass 2: the bench lines on the SAME binary as profiles/measure_r05.sh, with profiles/pmc_traffic.json of that pass in the
# tree (bench.py reads traffic / instruction counts / class mix from it and marks them stale when the build ids differ).
set -u
O=gpurun_out/r05m
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python bench.py --workload depth --steps 200 --warmup 20 > $O/bench_depth.json 2>/dev/null
python bench.py --workload lidar --steps 100 --warmup 10 > $O/bench_lidar.json 2>/dev/null
python bench.py --workload lidar_velocity --steps 100 --warmup 10 > $O/bench_lidar_velocity.json 2>/dev/null
python bench.py --workload lidar_nav --steps 200 --warmup 20 > $O/bench_lidar_nav.json 2>/dev/null
AGX_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-lidar --no-strict > $O/bench_forced_dist_world1.json 2> $O/bench_forced_dist.err
python bench.py --gpus 1 --exchange-selftest-only > $O/exchange_selftest_world1.jsonl 2> $O/exchange_selftest.err
PYTHONPATH=. python examples/benchmark.py --steps 5000 > $O/reference_benchmark_recipe.txt 2>/dev/null
PYTHONPATH=. python examples/benchmark.py --rendering --steps 2000 >> $O/reference_benchmark_recipe.txt 2>/dev/null
PYTHONPATH=. python examples/benchmark.py --num-envs 8192 --steps 5000 >> $O/reference_benchmark_recipe.txt 2>/dev/null
python profiles/small_batch_r02.py > /dev/null 2>&1; cp gpurun_out/r02_small_batch.txt $O/small_batch.txt 2>/dev/null
python - <<'P'
import json
for f in ("bench_default", "bench_driver_style", "bench_depth", "bench_lidar", "bench_lidar_velocity", "bench_lidar_nav", "bench_forced_dist_world1"):
    try:
        d = json.loads(open("gpurun_out/r05m/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"].get("traffic_stale"), d["roofline"].get("counters_stale"))
    except Exception as e:
        print(f, "unreadable", e)
P
cat $O/exchange_selftest_world1.jsonl | cut -c1-200
