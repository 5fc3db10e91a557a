set -u
O=gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 profiles/src/valu_peak.hip -o $O/valu_peak 2>/dev/null
$O/valu_peak > $O/valu_peak.jsonl
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/valu_peak_pmc -o p -- $O/valu_peak > $O/valu_peak_pmc.jsonl 2> $O/valu_peak_pmc.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o p -- python bench.py --no-cpu-baseline > $O/bench_default_under_rocprofv3.json 2> $O/prof_default.err
echo "prof rc=$?"
find $O -name "*kernel_trace.csv" -size +20M -delete
rm -f $O/valu_peak
cat $O/valu_peak.jsonl
ls $O $O/prof_default $O/valu_peak_pmc
