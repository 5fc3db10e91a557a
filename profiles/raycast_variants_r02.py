"""Round-2 ray-cast experiments, each behind a compile-time switch and kept only if
tests/test_gpu_raycast.py stays bit-exact:
   tile16x4     AGX_RAY_TILE_W=16   16 x 4 pixel tiles (64-byte row segments instead of 32)
   stereo8      AGX_RAY_STEREO_WAVES=8   stereo variant forced under 64 VGPRs (8 waves per SIMD)
   python profiles/raycast_variants_r02.py build        (here)
   python profiles/raycast_variants_r02.py run          (GPU box) -> gpurun_out/r02_raycast_variants.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {"default": [], "tile16x4": ["-DAGX_RAY_TILE_W=16"], "stereo8": ["-DAGX_RAY_STEREO_WAVES=8"]}


def lib(tag):
    return os.path.join(ROOT, "aerial_gym_simulator_amd", "lib", f"libagx_var_ray_{tag}.so")


if sys.argv[1] == "build":
    from aerial_gym_simulator_amd import _build

    for tag, flags in VARIANTS.items():
        print(tag, _build.build_library(extra_flags=flags, lib_path=lib(tag)))
else:
    out = []
    for tag in VARIANTS:
        env = dict(os.environ, AGX_LIB_PATH=lib(tag))
        t = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_raycast.py"), "-q", "-x"], env=env,
                           capture_output=True, text=True)
        verdict = t.stdout.strip().splitlines()[-1] if t.stdout.strip() else "no output"
        row = {"variant": tag, "bit_exact_tests": verdict}
        for wl, n in (("depth", 8192), ("lidar", 4096)):
            b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--num-envs", str(n), "--steps", "60", "--warmup", "6"],
                               env=env, capture_output=True, text=True).stdout.strip().splitlines()
            d = json.loads(b[-1])
            row[wl] = {"ms_per_step": d["ms_per_step"], "raycast_us": d["roofline"]["launch_us"], "rays_per_s": d["roofline"]["rays_per_s"]}
        if tag in ("default", "stereo8"):
            s = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "raycast_variants_f1.py"), "2048"], env=env, capture_output=True, text=True)
            row["f1_variants"] = [l for l in s.stdout.splitlines() if "us" in l or "ms" in l][-8:]
        out.append(row)
        print(json.dumps(row), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "r02_raycast_variants.txt"), "w") as f:
        for r in out:
            f.write(json.dumps(r) + "\n")
