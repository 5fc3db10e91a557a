#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_task_trace.py tests/test_gpu_full_size_parity.py::test_config1_every_env_of_8192_for_50_steps tests/test_gpu_world2_exchange.py tests/test_gpu_exchange.py tests/test_gpu_examples.py tests/test_gpu_lean_step.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -n "^FAILED\|^E  " $O/pytest.log | head -10
for f in 0 1; do AGX_SINGLE_LAUNCH=$f timeout 300 python bench.py --no-depth --no-cpu-baseline > $O/bench_single$f.json 2> $O/bench_single$f.err; done
python - <<'PY'
import json
for f in (0,1):
    d=json.loads(open(f"gpurun_out/r04i/bench_single{f}.json").read().strip().splitlines()[-1])
    print("single",f, d["value"], d["ms_per_step"], d["timed_regions"]["ms_per_step"], d["single_launch_steps"]["count"], d["single_launch_steps"]["of"])
PY
for f in 0 1; do AGX_SINGLE_LAUNCH=$f timeout 300 python bench.py --no-depth --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style single',$f, d['value'], d['timed_regions']['ms_per_step'], d['single_launch_steps']['count'], d['single_launch_steps']['of'])"; done
AGX_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-depth 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forced dist', d['value'], {k:(v.get('value'),v.get('ms_per_step')) for k,v in d['exchange'].items() if isinstance(v,dict) and 'value' in v})"
