"""BVH key experiments: build variants with -DAGX_BVH_KEY=k (+ ray stats), then measure node visits and frame time.
    python profiles/bvhkeys.py build ; (GPU) python profiles/bvhkeys.py run"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBS = {k: os.path.join(ROOT, "aerial_gym_simulator_amd", "lib", f"libagx_var_bvhkey{k}.so") for k in (0, 1)}  # k = AGX_BVH_SUBKEY
if sys.argv[1] == "build":
    from aerial_gym_simulator_amd import _build
    for k, path in LIBS.items():
        print(_build.build_library(extra_flags=["-DAGX_RAY_STATS", f"-DAGX_BVH_SUBKEY={k}"], lib_path=path))
elif sys.argv[1] == "run":
    import subprocess
    for k in LIBS:
        for wl in ("depth", "depth@curriculum", "lidar_nav"):
            subprocess.run([sys.executable, __file__, "one", str(k), wl])
else:
    k, wl = int(sys.argv[2]), sys.argv[3]
    os.environ["AGX_LIB_PATH"] = LIBS[k]
    import torch, bench
    n = 1024
    from aerial_gym_simulator_amd.env_manager.env_manager import EnvManager
    EnvManager.bvh_prims_per_object = 12
    t = bench.make_task(wl.split("@")[0], n, "cuda:0", False, obstacles="curriculum" if "@" in wl else "all"); t.reset()
    print("obstacles in env:", t.obs_dict["num_obstacles_in_env"], end="  ")
    lib = ctypes.CDLL(LIBS[k])
    out = (ctypes.c_ulonglong * 8)()
    a = torch.rand(n, 4, device="cuda:0") * 2 - 1
    for _ in range(5): t.step(a)
    torch.cuda.synchronize(); lib.agx_debug_ray_stats(out, 1)
    for _ in range(20): t.step(a)
    torch.cuda.synchronize(); lib.agx_debug_ray_stats(out, 1)
    pk, visits, leafs, lanes = out[0], out[1], out[2], out[3]
    kt = bench.kernel_time_raycast(t)
    print(f"key {k} {wl:17s}: node visits/packet {visits/pk:6.1f}  leaf tests/packet {leafs/pk:5.1f}  lanes/leaf {lanes/max(leafs,1):5.1f}  raycast {kt*1e3:.3f} ms (n={n}, stats build)")
