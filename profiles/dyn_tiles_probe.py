"""k_env_step<4, position, single, one lane per env> at 2^21 envs: the two-tile form of the kernel (profiles/src/
env_step_tiles_experiment_r04.patch on top of the buffer-access kernel: the next tile's core loads requested before the current
tile's arithmetic) at 4 / 3 / 2 waves per SIMD, against the shipped one-tile kernel at 4 waves.  Variant libraries are built by
`--build` from the patched source; each measurement runs in its own process (AGX_LIB_PATH selects the library).
    python profiles/dyn_tiles_probe.py --build ; python profiles/dyn_tiles_probe.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--child" in sys.argv:
    import torch

    import bench

    out = {"variant": os.environ.get("AGX_VARIANT", "default"), "AGX_DYN_TILES": os.environ.get("AGX_DYN_TILES")}
    for lean in (False, True):
        n = bench.LEAN_AT_SCALE_ENVS if lean else 1 << 21
        task = bench.make_task("dynamics", n, "cuda:0", False, lean=lean)
        task.reset()
        a = [torch.rand(n, 4, device="cuda:0") * 2 - 1]
        for _ in range(3):
            task.step(a[0])
        t, k = bench.kernel_time_dynamics(task, a, reps=30)
        out["lean" if lean else "all_tensors"] = {"in_step_us": round(t["in_step"] * 1e6, 2), "back_to_back_us": round(t["back_to_back"] * 1e6, 2)}
        del task
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)
    sys.exit(0)

from aerial_gym_simulator_amd import _build  # noqa: E402

variants = {"tiles_waves4": ["-DAGX_DYN_WAVES_LEAN_LAWS=4"], "tiles_waves3": ["-DAGX_DYN_WAVES_LEAN_LAWS=3"],
            "tiles_waves2": ["-DAGX_DYN_WAVES_LEAN_LAWS=2"]}
runs = [("default", None, None)]
for name, flags in variants.items():
    lib = os.path.join(ROOT, "aerial_gym_simulator_amd", "lib", "libaerialgym_hip_%s.so" % name)
    if "--build" in sys.argv:
        _build.build_library(extra_flags=flags, lib_path=lib)
        print("built", lib, flush=True)
        continue
    if os.path.exists(lib):
        runs += [(name, lib, "2"), (name, lib, "1")]
if "--build" not in sys.argv:
    for name, lib, tiles in runs:
        env = dict(os.environ, AGX_VARIANT=name)
        if lib:
            env["AGX_LIB_PATH"] = lib
        if tiles:
            env["AGX_DYN_TILES"] = tiles
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-500:], flush=True)
