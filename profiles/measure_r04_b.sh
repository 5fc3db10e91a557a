#!/bin/bash
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_raycast.py tests/test_gpu_warp_kernels.py -q -m gpu -x > gpurun_out/r04b/pytest_raycast.txt 2>&1; echo "rc=$?" >> gpurun_out/r04b/pytest_raycast.txt
tail -3 gpurun_out/r04b/pytest_raycast.txt
timeout 600 python profiles/raycast_split_probe.py depth > gpurun_out/r04b/split_depth.jsonl 2> gpurun_out/r04b/split_depth.err
timeout 600 python profiles/raycast_split_probe.py lidar > gpurun_out/r04b/split_lidar.jsonl 2> gpurun_out/r04b/split_lidar.err
cat gpurun_out/r04b/split_depth.jsonl gpurun_out/r04b/split_lidar.jsonl
tail -3 gpurun_out/r04b/split_depth.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r04b/counters_avail.txt 2>&1
grep -c . $GRAFT_REPO_ROOT/gpurun_out/r04b/counters_avail.txt
