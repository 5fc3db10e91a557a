#!/bin/bash
# Round 4, after the load-hoisting pass over the env-step / reset kernels (every load of a kernel requested in ONE round trip at its
# top): the GPU suite, then the default bench line twice (driver form and 5 x 2000 steps).
#   gpurun --timeout 900 -- 'bash profiles/measure_r04_l.sh'
set -u
O=gpurun_out/r04p
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python bench.py --workload lidar_nav --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_lidar_nav.json 2>/dev/null
python - <<'P'
import json
for f in ("bench_default", "bench_driver_style", "bench_lidar_nav"):
    try:
        d = json.loads(open("gpurun_out/r04p/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], d["ms_per_step"], d.get("launch_us_detail"))
    for k in ("plus_depth", "roofline_at_scale", "roofline_at_scale_all_tensors", "roofline", "roofline_reset_obs"):
        v = d.get(k)
        if isinstance(v, dict):
            print("  ", k, {a: v[a] for a in list(v)[:8]})
P
