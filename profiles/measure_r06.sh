#!/bin/bash
# Round-6 measurement pass 1 on the GPU box (one gpurun call): the whole GPU test suite, the counters bench.py's roofline keys are
# read from (six rocprofv3 PMC passes, --kernel-trace only next to --pmc, one counter group per run) and the per-kernel statistics of
# the bench commands.   gpurun --timeout 2400 -- 'bash profiles/measure_r06.sh'   then here:  bash profiles/collect_r06.sh pmc
# Pass 2 (same binary, counters now in the tree):   gpurun -- 'bash profiles/measure_r06_benchlines.sh'; bash profiles/collect_r06.sh
set -u
O=gpurun_out/r06m
mkdir -p $O
export TMPDIR=/tmp
export AGX_BUILD_ID_OUT=$PWD/$O/pmc_build_id.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o p -- python bench.py --no-cpu-baseline > $O/bench_default_under_rocprofv3.json 2> $O/prof_default.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python profiles/pmc_probe.py --nav > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python profiles/pmc_probe.py --nav > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq -o p -- python profiles/pmc_probe.py --nav > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq2 -o p -- python profiles/pmc_probe.py --nav > $O/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $O/pmc_sq3 -o p -- python profiles/pmc_probe.py --nav > $O/pmc_sq3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT --output-format csv -d $O/pmc_sq4 -o p -- python profiles/pmc_probe.py --nav > $O/pmc_sq4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_depth -o p -- python bench.py --workload depth --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/prof_depth.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lidar -o p -- python bench.py --workload lidar --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/prof_lidar.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lidar_nav -o p -- python bench.py --workload lidar_nav --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/prof_lidar_nav.err
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O; ls $O | head -40
tail -2 $O/*.log | cut -c1-160 | tail -24
