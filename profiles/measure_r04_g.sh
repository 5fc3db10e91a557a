#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_raycast.py tests/test_gpu_warp_kernels.py -q -m gpu -x > $O/pytest_raycast.txt 2>&1; echo "rc=$?" >> $O/pytest_raycast.txt; tail -2 $O/pytest_raycast.txt
timeout 600 python profiles/raycast_split_probe.py depth > $O/split_depth.jsonl 2> $O/split_depth.err
timeout 600 python profiles/raycast_split_probe.py lidar > $O/split_lidar.jsonl 2> $O/split_lidar.err
cat $O/split_depth.jsonl $O/split_lidar.jsonl
