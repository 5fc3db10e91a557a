"""Traversal statistics of k_raycast (experimental build with -DAGX_RAY_STATS).
    python profiles/raystats.py build ; (GPU) python profiles/raystats.py run [depth|lidar]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "aerial_gym_simulator_amd", "lib", "libagx_var_raystats.so")
if sys.argv[1] == "build":
    from aerial_gym_simulator_amd import _build
    print(_build.build_library(extra_flags=["-DAGX_RAY_STATS"], lib_path=LIB))
else:
    os.environ["AGX_LIB_PATH"] = LIB
    import torch, bench
    wl = sys.argv[2] if len(sys.argv) > 2 else "depth"
    n = 256
    t = bench.make_task(wl, n, "cuda:0", False); t.reset()
    lib = ctypes.CDLL(LIB)
    out = (ctypes.c_ulonglong * 8)()
    a = torch.rand(n, 4, device="cuda:0") * 2 - 1
    for _ in range(5): t.step(a)
    torch.cuda.synchronize(); lib.agx_debug_ray_stats(out, 1)
    K = 20
    for _ in range(K): t.step(a)
    torch.cuda.synchronize(); lib.agx_debug_ray_stats(out, 1)
    pk, visits, leafs, leaf_lanes = out[0], out[1], out[2], out[3]
    print(f"{wl}: packets {pk}  node visits/packet {visits/pk:.1f}  leaf tests/packet {leafs/pk:.1f}  active lanes per leaf test {leaf_lanes/max(leafs,1):.1f}")
