#!/bin/bash
# Round-4 measurement pass on the GPU box (one gpurun call): everything bench.py's roofline keys are read from, the bench lines, the
# per-kernel statistics of the same commands, and the whole GPU test suite.
#   gpurun --timeout 2700 -- 'bash profiles/measure_r04.sh'
# Afterwards, in the build container:  bash profiles/collect_r04.sh
# PMC passes carry --kernel-trace only (no sys / hip / hsa tracing next to counters), one counter group per run.
set -u
O=gpurun_out/r04m
mkdir -p $O
export TMPDIR=/tmp
export AGX_BUILD_ID_OUT=$PWD/$O/pmc_build_id.txt
timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
hipcc --offload-arch=gfx950 -O3 profiles/src/valu_peak.hip -o $O/valu_peak 2>/dev/null
$O/valu_peak > $O/valu_peak.jsonl
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/valu_peak_pmc -o p -- $O/valu_peak > $O/valu_peak_pmc.jsonl 2> $O/valu_peak_pmc.err
# the default bench line, un-profiled, the driver's form of it, and under the kernel trace (same command: the per-kernel averages the line must agree with)
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o p -- python bench.py --no-cpu-baseline > $O/bench_default_under_rocprofv3.json 2> $O/prof_default.err
# counters: separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python profiles/pmc_probe.py --nav > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python profiles/pmc_probe.py --nav > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq -o p -- python profiles/pmc_probe.py --nav > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq2 -o p -- python profiles/pmc_probe.py --nav > $O/pmc_sq2.log 2>&1
# instruction classes of the vector instructions (what the pipe is busy with: bench.py valu.pipe_busy), two more passes
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $O/pmc_sq3 -o p -- python profiles/pmc_probe.py --nav > $O/pmc_sq3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT --output-format csv -d $O/pmc_sq4 -o p -- python profiles/pmc_probe.py --nav > $O/pmc_sq4.log 2>&1
# the other workloads as their own bench lines (+ kernel statistics of the LiDAR one)
python bench.py --workload depth --steps 200 --warmup 20 > $O/bench_depth.json 2>/dev/null
python bench.py --workload lidar --steps 100 --warmup 10 > $O/bench_lidar.json 2>/dev/null
python bench.py --workload lidar_velocity --steps 100 --warmup 10 > $O/bench_lidar_velocity.json 2>/dev/null
python bench.py --workload lidar_nav --steps 200 --warmup 20 > $O/bench_lidar_nav.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_depth -o p -- python bench.py --workload depth --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/prof_depth.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lidar -o p -- python bench.py --workload lidar --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/prof_lidar.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lidar_nav -o p -- python bench.py --workload lidar_nav --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/prof_lidar_nav.err
AGX_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline > $O/bench_forced_dist_world1.json 2> $O/bench_forced_dist.err
PYTHONPATH=. python examples/benchmark.py --steps 5000 > $O/reference_benchmark_recipe.txt 2>/dev/null
PYTHONPATH=. python examples/benchmark.py --rendering --steps 2000 >> $O/reference_benchmark_recipe.txt 2>/dev/null
PYTHONPATH=. python examples/benchmark.py --num-envs 8192 --steps 5000 >> $O/reference_benchmark_recipe.txt 2>/dev/null
python profiles/small_batch_r02.py > /dev/null 2>&1; cp gpurun_out/r02_small_batch.txt $O/small_batch.txt 2>/dev/null
# keep what is needed, drop the bulky traces
find $O -name "*kernel_trace.csv" -size +20M -delete
rm -f $O/valu_peak
du -sh $O; ls $O | head -60
tail -2 $O/*.log | cut -c1-200 | tail -30
