#!/bin/bash
# round 4, first GPU call: the new parity tests (reference-executed warp goldens, full-size configs, bench launcher)
mkdir -p gpurun_out/r04a
python -c "import os; print('cores', os.cpu_count())" > gpurun_out/r04a/host.txt
timeout 1500 python -m pytest tests/test_gpu_warp_kernels.py tests/test_bench_launcher.py tests/test_gpu_full_size_parity.py -q -s -m gpu -x > gpurun_out/r04a/pytest_new.txt 2>&1
echo "rc=$?" >> gpurun_out/r04a/pytest_new.txt
tail -30 gpurun_out/r04a/pytest_new.txt
