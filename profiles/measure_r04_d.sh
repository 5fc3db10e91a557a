#!/bin/bash
# round 4, mid-round check: the whole GPU suite on the current tree + the default bench line
O=gpurun_out/r04d; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04d/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","timed_regions")})
for k in ("roofline","roofline_reset_obs","roofline_step"):
    print(k, {a:b for a,b in d.get(k,{}).items() if a in ("bound","achieved","peak","frac","traffic","launch_us","kernel_us_sum","share_env_step","frac_over_step_time")})
p=d["plus_depth"]; print("plus_depth", p["value"], p["ms_per_step"], p["raycast_launch_us"], {a:b for a,b in p["raycast_roofline"].items() if a in ("bound","achieved","peak","frac","counters_stale","bound_note")})
print(d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("kind"))
PY
