#!/bin/bash
# Round 6, pass 2: the bench lines on the SAME binary as profiles/measure_r06.sh, with profiles/pmc_traffic.json of that pass in the
# tree (bench.py reads traffic / instruction counts / class mix from it and marks them stale when the build ids differ).
# Since round 6 the stdout line is the compact one (< 6 KB); the full record of each run is gpurun_out/bench_detail.json, copied next
# to the line as <name>_detail.json.
set -u
O=gpurun_out/r06m
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; python bench.py "$@" > $O/$name.json 2> $O/$name.err; cp gpurun_out/bench_detail.json $O/${name}_detail.json 2>/dev/null; }
run bench_default
run bench_driver_style --steps 20 --warmup 5
run bench_depth --workload depth --steps 200 --warmup 20
run bench_lidar --workload lidar --steps 100 --warmup 10
run bench_lidar_velocity --workload lidar_velocity --steps 100 --warmup 10
run bench_lidar_nav --workload lidar_nav --steps 200 --warmup 20
AGX_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-lidar --no-strict > $O/bench_forced_dist_world1.json 2> $O/bench_forced_dist.err
cp gpurun_out/bench_detail.json $O/bench_forced_dist_world1_detail.json 2>/dev/null
python bench.py --gpus 1 --exchange-selftest-only > $O/exchange_selftest_world1.jsonl 2> $O/exchange_selftest.err
PYTHONPATH=. python examples/benchmark.py --steps 5000 > $O/reference_benchmark_recipe.txt 2>/dev/null
PYTHONPATH=. python examples/benchmark.py --rendering --steps 2000 >> $O/reference_benchmark_recipe.txt 2>/dev/null
PYTHONPATH=. python examples/benchmark.py --num-envs 8192 --steps 5000 >> $O/reference_benchmark_recipe.txt 2>/dev/null
python profiles/small_batch_r02.py > /dev/null 2>&1; cp gpurun_out/r02_small_batch.txt $O/small_batch.txt 2>/dev/null
python profiles/strict_probe_r06.py > $O/strict_probe_after.jsonl 2>/dev/null
python profiles/scene_phase_probe_r06.py 2>/dev/null | tail -1 > $O/scene_refresh_phases.json
python - <<'P'
import json
for f in ("bench_default", "bench_driver_style", "bench_depth", "bench_lidar", "bench_lidar_velocity", "bench_lidar_nav", "bench_forced_dist_world1"):
    try:
        raw = open("gpurun_out/r06m/%s.json" % f).read().strip().splitlines()[-1]
        d = json.loads(raw)
        print(f, len(raw), "bytes", d["value"], d["ms_per_step"], d["roofline"].get("traffic_stale"), d.get("value_strict_rng"))
    except Exception as e:
        print(f, "unreadable", e)
P
cat $O/exchange_selftest_world1.jsonl | cut -c1-200
