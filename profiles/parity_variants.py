"""Arithmetic variants of the dynamics kernel against the parity gates AND the clock (run on the GPU box).

    python profiles/parity_variants.py build      # here: cross-compile the variant libraries
    python profiles/parity_variants.py run        # GPU box: parity maxima (soft gates) + kernel / step time per variant

Variants: hardware rcp / rsq / sqrt (+ Newton) vs correctly rounded division / sqrt, fma contraction on / off."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {
    "fast_rcp+contract": [],
    "exact_div+contract": ["-DAGX_DYN_FAST_RCP=0"],
    "exact_div+no_contract": ["-DAGX_DYN_FAST_RCP=0", "-DAGX_DYN_CONTRACT=0"],
    "fast_rcp+no_contract": ["-DAGX_DYN_CONTRACT=0"],
}


def lib(tag):
    return os.path.join(ROOT, "aerial_gym_simulator_amd", "lib", f"libagx_var_{tag.replace('+', '_')}.so")


if sys.argv[1] == "build":
    from aerial_gym_simulator_amd import _build

    for tag, flags in VARIANTS.items():
        print(tag, _build.build_library(extra_flags=flags, lib_path=lib(tag)))
else:
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    summary = {}
    for tag in VARIANTS:
        rep = os.path.join(out_dir, f"parity_{tag.replace('+', '_')}.json")
        env = dict(os.environ, AGX_LIB_PATH=lib(tag), AGX_PARITY_SOFT="1", AGX_PARITY_REPORT=rep)
        subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_dynamics.py"),
                        os.path.join(ROOT, "tests", "test_gpu_task_trace.py"), "-q", "-k",
                        "single_substep or fused_k or config1_trace"], env=env, capture_output=True, text=True)
        rows = json.load(open(rep))
        worst = {}
        for k, v in rows.items():
            key = k.split("[")[0] + ("/" + k.split("/")[-1] if "/" in k else "")
            worst[key] = max(worst.get(key, 0.0), v["max"])
        b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1500", "--warmup", "150", "--no-cpu-baseline",
                            "--no-depth"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(b)
        summary[tag] = {"parity_worst": worst, "us_per_step": 1e3 * d["ms_per_step"], "kernel_us_8192": d["roofline"]["launch_us"],
                        "kernel_us_2M": d.get("roofline_at_scale", {}).get("launch_us")}
        print(tag, json.dumps(summary[tag]), flush=True)
    json.dump(summary, open(os.path.join(out_dir, "parity_variants.json"), "w"), indent=1, sort_keys=True)
