# final GPU pass of round 3: the whole GPU test suite, then the default bench line, the same under rocprofv3, and the
# bench with the sharded code path forced in a world of one
set -u
O=gpurun_out/r03f
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
cp profiles/r03_parity_report.json $O/parity_report_before.json 2>/dev/null
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"
AGX_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline > $O/bench_forced_dist_world1.json 2> $O/bench_forced_dist.err
echo "forced dist rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o p -- python bench.py --no-cpu-baseline > $O/bench_default_under_rocprofv3.json 2> $O/prof_default.err
echo "prof rc=$?"
find $O -name "*kernel_trace.csv" -size +20M -delete
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
ls $O $O/prof_default
