#!/bin/bash
# Round-3 measurement pass on the GPU box (one gpurun call): everything bench.py's roofline keys are read from.
#   gpurun --timeout 1500 -- 'bash profiles/measure_r03.sh'
# Afterwards, in the build container:
#   python profiles/collect_valu_peak.py gpurun_out/r03m/valu_peak.jsonl gpurun_out/r03m/valu_peak_pmc/p_counter_collection.csv gpurun_out/r03m/valu_peak_pmc.jsonl
#   python profiles/collect_pmc.py gpurun_out/r03m/pmc_fetch/p_counter_collection.csv gpurun_out/r03m/pmc_write/p_counter_collection.csv \
#          gpurun_out/r03m/pmc_sq/p_counter_collection.csv gpurun_out/r03m/pmc_build_id.txt gpurun_out/r03m/pmc_sq2/p_counter_collection.csv
# PMC passes carry --kernel-trace only (no sys / hip / hsa tracing next to counters).
set -u
O=gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp
export AGX_BUILD_ID_OUT=$PWD/$O/pmc_build_id.txt
hipcc --offload-arch=gfx950 -O3 profiles/src/valu_peak.hip -o $O/valu_peak 2>/dev/null
$O/valu_peak > $O/valu_peak.jsonl
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/valu_peak_pmc -o p -- $O/valu_peak > $O/valu_peak_pmc.jsonl 2> $O/valu_peak_pmc.err
# the default bench line, un-profiled and under the kernel trace (same command: the per-kernel averages the line must agree with)
python bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o p -- python bench.py --no-cpu-baseline > $O/bench_default_under_rocprofv3.json 2> $O/prof_default.err
# counters: separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python profiles/pmc_probe.py --nav > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python profiles/pmc_probe.py --nav > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq -o p -- python profiles/pmc_probe.py --nav > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq2 -o p -- python profiles/pmc_probe.py --nav > $O/pmc_sq2.log 2>&1
# keep what is needed, drop the bulky traces
find $O -name "*kernel_trace.csv" -size +20M -delete
rm -f $O/valu_peak
ls -la $O $O/*/ | head -60
tail -3 $O/*.log | cut -c1-200
