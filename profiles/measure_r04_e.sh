#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_task_trace.py tests/test_gpu_full_size_parity.py tests/test_gpu_world2_exchange.py tests/test_gpu_exchange.py tests/test_gpu_api_surface.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -n "^FAILED\|^E  " $O/pytest.log | head -10
for f in 0 1; do AGX_FUSE_OBS=$f timeout 300 python bench.py --no-depth --no-cpu-baseline > $O/bench_fuse$f.json 2> $O/bench_fuse$f.err; done
python - <<'PY'
import json
for f in (0,1):
    d=json.loads(open(f"gpurun_out/r04e/bench_fuse{f}.json").read().strip().splitlines()[-1])
    print("fuse",f, d["value"], d["ms_per_step"], d["timed_regions"]["ms_per_step"], d["roofline"]["launch_us"], d.get("roofline_reset_obs",{}).get("launch_us"))
PY
for f in 0 1; do AGX_FUSE_OBS=$f timeout 300 python bench.py --no-depth --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style fuse',$f, d['value'], d['timed_regions']['ms_per_step'])"; done
