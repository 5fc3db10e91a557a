#!/bin/bash
# build container, after the gpurun passes of profiles/measure_r06*.sh: gpurun_out/r06m -> the committed profiles/r06_* files
#   bash profiles/collect_r06.sh pmc    (after pass 1: only the counter file, so that pass 2 runs against it)
#   bash profiles/collect_r06.sh        (after pass 2: everything)
set -u
O=gpurun_out/r06m
python profiles/collect_pmc.py $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv $O/pmc_sq/p_counter_collection.csv $O/pmc_build_id.txt $O/pmc_sq2/p_counter_collection.csv $O/pmc_sq3/p_counter_collection.csv $O/pmc_sq4/p_counter_collection.csv > /dev/null
[ "${1:-}" = "pmc" ] && { python -c "import json; d=json.load(open('profiles/pmc_traffic.json')); print('build', d['build_id'], [k for k in d if k.startswith('k_raycast') and not k.endswith('detail')])"; exit 0; }
for f in bench_default bench_driver_style bench_default_under_rocprofv3 bench_depth bench_lidar bench_lidar_velocity bench_lidar_nav bench_forced_dist_world1; do
  [ -s $O/$f.json ] && tail -1 $O/$f.json | python -m json.tool > profiles/r06_$f.json
done
for f in bench_default bench_driver_style bench_depth bench_lidar; do  # the full records behind the compact lines
  [ -s $O/${f}_detail.json ] && cp $O/${f}_detail.json profiles/r06_${f}_detail.json
done
cp $O/strict_probe_after.jsonl profiles/r06_strict_probe_after.jsonl 2>/dev/null
cp $O/scene_refresh_phases.json profiles/r06_scene_refresh_phases.json 2>/dev/null
cp $O/prof_default/p_kernel_stats.csv profiles/r06_bench_default_kernel_stats.csv 2>/dev/null
cp $O/prof_depth/p_kernel_stats.csv profiles/r06_bench_depth_kernel_stats.csv 2>/dev/null
cp $O/prof_lidar/p_kernel_stats.csv profiles/r06_bench_lidar_kernel_stats.csv 2>/dev/null
cp $O/prof_lidar_nav/p_kernel_stats.csv profiles/r06_bench_lidar_nav_kernel_stats.csv 2>/dev/null
cp $O/parity_report.json profiles/r06_parity_report.json 2>/dev/null
cp $O/reference_benchmark_recipe.txt profiles/r06_reference_benchmark_recipe.txt 2>/dev/null
cp $O/small_batch.txt profiles/r06_small_batch.txt 2>/dev/null
cp $O/exchange_selftest_world1.jsonl profiles/r06_exchange_selftest_world1.jsonl 2>/dev/null
grep -E "passed|failed|skipped" $O/pytest_gpu.log | tail -3 > profiles/r06_pytest_gpu_summary.txt
ls -la profiles/r06_*
