"""k_env_step<4, position, single, !WIDE> at 2^21 envs: waves per SIMD the kernel is compiled for (AGX_DYN_WAVES).
Builds variant libraries next to the default one and times the env-step launch of each in its own process.
    python profiles/dyn_waves_probe.py            (driver)
    python profiles/dyn_waves_probe.py --child    (one measurement; AGX_LIB_PATH selects the library)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--child" in sys.argv:
    import torch

    import bench

    out = {"variant": os.environ.get("AGX_VARIANT", "default")}
    for lean in (False, True):
        n = bench.LEAN_AT_SCALE_ENVS if lean else 1 << 21
        task = bench.make_task("dynamics", n, "cuda:0", False, lean=lean)
        task.reset()
        a = [torch.rand(n, 4, device="cuda:0") * 2 - 1]
        for _ in range(3):
            task.step(a[0])
        t, k = bench.kernel_time_dynamics(task, a, reps=30)
        out["lean" if lean else "default"] = {"in_step_us": t["in_step"] * 1e6, "back_to_back_us": t["back_to_back"] * 1e6}
        del task
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)
    sys.exit(0)

from aerial_gym_simulator_amd import _build  # noqa: E402

variants = {"default": None, "waves4": ["-DAGX_DYN_WAVES=4"], "waves2": ["-DAGX_DYN_WAVES=2"]}
for name, flags in variants.items():
    env = dict(os.environ, AGX_VARIANT=name)
    if flags is not None:
        lib = os.path.join(ROOT, "aerial_gym_simulator_amd", "lib", "libaerialgym_hip_%s.so" % name)
        if not os.path.exists(lib):
            _build.build_library(extra_flags=flags, lib_path=lib)
        env["AGX_LIB_PATH"] = lib
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-500:], flush=True)
