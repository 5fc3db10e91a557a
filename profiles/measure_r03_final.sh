# final GPU pass of round 3 (one gpurun call, ~15 min): the whole GPU test suite, every measurement bench.py's keys are read
# from (profiles/measure_r03.sh), the sharded code path forced in a world of one, the reference's benchmark recipe
#   gpurun --timeout 2400 -- 'bash profiles/measure_r03_final.sh'
set -u
O=gpurun_out/r03f
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
bash profiles/measure_r03.sh > $O/measure_r03.log 2>&1
echo "measure rc=$?"
AGX_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline > $O/bench_forced_dist_world1.json 2> $O/bench_forced_dist.err
echo "forced dist rc=$?"
PYTHONPATH=. python examples/benchmark.py --steps 5000 > $O/reference_benchmark_recipe.txt 2>/dev/null
PYTHONPATH=. python examples/benchmark.py --rendering --steps 2000 >> $O/reference_benchmark_recipe.txt 2>/dev/null
PYTHONPATH=. python examples/benchmark.py --num-envs 8192 --steps 5000 >> $O/reference_benchmark_recipe.txt 2>/dev/null
python profiles/small_batch_r02.py > /dev/null 2>&1; cp gpurun_out/r02_small_batch.txt $O/small_batch.txt 2>/dev/null
cat $O/reference_benchmark_recipe.txt
ls $O gpurun_out/r03m | head -60
