#!/bin/bash
# Round measurement pass on the GPU box (run through gpurun from the repo root):
#   1. three rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU+SQ_WAVES; separate runs as the MI355X guide prescribes)
#   2. rocprofv3 --kernel-trace --stats of the default bench command and of the depth / LiDAR workloads
#   3. the bench lines themselves (no profiler attached)
# Outputs under gpurun_out/rNN_*; profiles/collect_pmc.py and a copy step turn them into the committed files.
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_WAVES"; do
  set -- $pass; name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${R}_pmc_$name -o p -- python $GRAFT_REPO_ROOT/profiles/pmc_probe.py --nav > $OUT/${R}_pmc_$name.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${R}_prof_default -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $OUT/${R}_bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${R}_prof_lidar -o p -- python $GRAFT_REPO_ROOT/bench.py --workload lidar --steps 60 --warmup 6 > $OUT/${R}_bench_lidar_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > $OUT/${R}_bench_default.json 2> $OUT/${R}_bench_default.err
timeout 300 python bench.py --workload depth --steps 200 --warmup 20 > $OUT/${R}_bench_depth.json 2>/dev/null
timeout 300 python bench.py --workload lidar --steps 100 --warmup 10 > $OUT/${R}_bench_lidar.json 2>/dev/null
timeout 300 python bench.py --workload lidar_nav --steps 200 --warmup 20 > $OUT/${R}_bench_lidar_nav.json 2>/dev/null
ls $OUT | grep ${R}_ | head -40
