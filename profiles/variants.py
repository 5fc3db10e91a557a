"""Build-variant A/B for k_env_step (run here: builds; run on the GPU box: times them).
    python profiles/variants.py build          # cross-compile variants into aerial_gym_simulator_amd/lib/
    python profiles/variants.py time           # on the GPU: bench each variant in a fresh process
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RAY_VARIANTS = {
    "ray512lds": ["-DAGX_RAY_THREADS=512", "-DAGX_RAY_USE_LDS=1"],
    "ray1024lds": ["-DAGX_RAY_THREADS=1024", "-DAGX_RAY_USE_LDS=1"],
    "ray128glb": ["-DAGX_RAY_THREADS=128"],
    "ray256glb": ["-DAGX_RAY_THREADS=256"],
    "ray512glb": ["-DAGX_RAY_THREADS=512"],
}
VARIANTS = {
    "c1w2": ["-DAGX_DYN_CONTRACT=1", "-DAGX_DYN_WAVES=2"],
    "c0w2": ["-DAGX_DYN_CONTRACT=0", "-DAGX_DYN_WAVES=2"],
    "c1w1": ["-DAGX_DYN_CONTRACT=1", "-DAGX_DYN_WAVES=1"],
    "c0w1": ["-DAGX_DYN_CONTRACT=0", "-DAGX_DYN_WAVES=1"],
    "c1w3": ["-DAGX_DYN_CONTRACT=1", "-DAGX_DYN_WAVES=3"],
}


def lib(tag):
    return os.path.join(ROOT, "aerial_gym_simulator_amd", "lib", f"libagx_var_{tag}.so")


if sys.argv[1] == "buildray":
    from aerial_gym_simulator_amd import _build

    for tag, flags in RAY_VARIANTS.items():
        print(tag, _build.build_library(extra_flags=flags, lib_path=lib(tag)))
elif sys.argv[1] == "timeray":
    for tag in RAY_VARIANTS:
        env = dict(os.environ, AGX_LIB_PATH=lib(tag))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "depth", "--steps", "60", "--warmup", "6"],
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print(tag, "depth env-steps/s %.2fM ms/step %.3f" % (d["value"] / 1e6, d["ms_per_step"]))
elif sys.argv[1] == "build":
    from aerial_gym_simulator_amd import _build

    for tag, flags in VARIANTS.items():
        print(tag, _build.build_library(extra_flags=flags, lib_path=lib(tag)))
else:
    for tag in VARIANTS:
        env = dict(os.environ, AGX_LIB_PATH=lib(tag))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1500", "--warmup", "150", "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print(tag, "env-steps/s %.1fM" % (d["value"] / 1e6), "us/step %.2f" % (1e3 * d["ms_per_step"]),
              "kernel_us %.2f" % d["roofline"]["launch_us"], "at-scale us %.1f frac %.3f" % (d["roofline_at_scale"]["launch_us"], d["roofline_at_scale"]["frac"]))
