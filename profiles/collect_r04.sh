#!/bin/bash
# build container, after `gpurun -- bash profiles/measure_r04.sh`: turn gpurun_out/r04m into the committed profiles/r04_* files
set -u
O=gpurun_out/r04m
python profiles/collect_valu_peak.py $O/valu_peak.jsonl $O/valu_peak_pmc/p_counter_collection.csv $O/valu_peak_pmc.jsonl
python profiles/collect_pmc.py $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv $O/pmc_sq/p_counter_collection.csv $O/pmc_build_id.txt $O/pmc_sq2/p_counter_collection.csv $O/pmc_sq3/p_counter_collection.csv $O/pmc_sq4/p_counter_collection.csv > /dev/null
for f in bench_default bench_driver_style bench_default_under_rocprofv3 bench_depth bench_lidar bench_lidar_velocity bench_lidar_nav bench_forced_dist_world1; do
  [ -s $O/$f.json ] && tail -1 $O/$f.json | python -m json.tool > profiles/r04_$f.json
done
cp $O/prof_default/p_kernel_stats.csv profiles/r04_bench_default_kernel_stats.csv 2>/dev/null
cp $O/prof_depth/p_kernel_stats.csv profiles/r04_bench_depth_kernel_stats.csv 2>/dev/null
cp $O/prof_lidar/p_kernel_stats.csv profiles/r04_bench_lidar_kernel_stats.csv 2>/dev/null
cp $O/prof_lidar_nav/p_kernel_stats.csv profiles/r04_bench_lidar_nav_kernel_stats.csv 2>/dev/null
cp $O/parity_report.json profiles/r04_parity_report.json 2>/dev/null
cp $O/reference_benchmark_recipe.txt profiles/r04_reference_benchmark_recipe.txt 2>/dev/null
cp $O/small_batch.txt profiles/r04_small_batch.txt 2>/dev/null
grep -E "passed|failed|skipped" $O/pytest_gpu.log | tail -3 > profiles/r04_pytest_gpu_summary.txt
ls -la profiles/r04_*
