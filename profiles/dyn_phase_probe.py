"""k_env_step<4, position, single, one lane per env> at 2^21 envs: how long its MEMORY phases take on their own.
Times the env-step launch of the default library and of a variant built from the source with
profiles/src/dyn_phase_probe_r04.patch applied and -DAGX_DYN_PHASE_PROBE=1 (every load and store of the kernel, no arithmetic
between them; the patch is not part of the product source), each in its own process -- the bound a perfect overlap of a wave's
load, arithmetic and store phases could reach from the memory side.
    git apply profiles/src/dyn_phase_probe_r04.patch && python profiles/dyn_phase_probe.py --build && git checkout aerial_gym_simulator_amd/csrc/agx_dynamics.hip
    python profiles/dyn_phase_probe.py            (driver)
    python profiles/dyn_phase_probe.py --child    (one measurement; AGX_LIB_PATH selects the library)"""
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
        out["lean" if lean else "all_tensors"] = {"in_step_us": t["in_step"] * 1e6, "back_to_back_us": t["back_to_back"] * 1e6}
        del task
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)
    sys.exit(0)

from aerial_gym_simulator_amd import _build  # noqa: E402

variants = {"default": None, "memory_phases_only": ["-DAGX_DYN_PHASE_PROBE=1"]}
for name, flags in variants.items():
    env = dict(os.environ, AGX_VARIANT=name)
    if flags is not None:
        lib = os.path.join(ROOT, "aerial_gym_simulator_amd", "lib", "libaerialgym_hip_%s.so" % name)
        if "--build" in sys.argv:
            _build.build_library(extra_flags=flags, lib_path=lib)
            continue
        if not os.path.exists(lib):
            print("variant library missing: see the module docstring", flush=True)
            continue
        env["AGX_LIB_PATH"] = lib
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-500:], flush=True)
