"""Experiment: how much would a SAH top-level order (instead of the Morton order of object centres) save?
Object codes = root-to-leaf path bits of a host-side (numpy) full-sweep SAH over the objects' AABBs, handed to an
experimental build of k_bvh_build (-DAGX_BVH_EXPERIMENT -DAGX_RAY_STATS).
    python profiles/bvh_sah_experiment.py build ; (GPU) python profiles/bvh_sah_experiment.py run"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "aerial_gym_simulator_amd", "lib", "libagx_var_bvhsah.so")


def sah_codes(lo, hi):
    """lo, hi [K,3] object AABBs -> uint32 path codes (bit 31 first), full-sweep SAH, leaves = single objects"""
    K = lo.shape[0]
    codes = np.zeros(K, np.uint32)

    def area(l, h):
        d = np.maximum(h - l, 0)
        return 2 * (d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0])

    def rec(idx, depth):
        if len(idx) <= 1 or depth >= 30:
            return
        best = (np.inf, None, None)
        cen = 0.5 * (lo[idx] + hi[idx])
        for ax in range(3):
            order = idx[np.argsort(cen[:, ax], kind="stable")]
            l_lo = np.minimum.accumulate(lo[order], axis=0); l_hi = np.maximum.accumulate(hi[order], axis=0)
            r_lo = np.minimum.accumulate(lo[order][::-1], axis=0)[::-1]; r_hi = np.maximum.accumulate(hi[order][::-1], axis=0)[::-1]
            n = len(order)
            k = np.arange(1, n)
            cost = area(l_lo[:-1], l_hi[:-1]) * k + area(r_lo[1:], r_hi[1:]) * (n - k)
            j = int(np.argmin(cost))
            if cost[j] < best[0]:
                best = (cost[j], order, j + 1)
        _, order, cut = best
        left, right = order[:cut], order[cut:]
        codes[right] |= np.uint32(1 << (31 - depth))
        rec(left, depth + 1); rec(right, depth + 1)

    rec(np.arange(K), 0)
    return codes


if sys.argv[1] == "build":
    from aerial_gym_simulator_amd import _build
    print(_build.build_library(extra_flags=["-DAGX_RAY_STATS", "-DAGX_BVH_EXPERIMENT"], lib_path=LIB))
else:
    os.environ["AGX_LIB_PATH"] = LIB
    import torch, bench
    from aerial_gym_simulator_amd import _lib
    for wl, obstacles in (("depth", "all"), ("depth", "curriculum"), ("lidar_nav", "all")):
        n = 256
        t = bench.make_task(wl, n, "cuda:0", False, obstacles=obstacles); t.reset()
        env = t.sim_env; sc = env.scene; sen = env.robot_manager.warp_sensor
        lib = ctypes.CDLL(LIB)
        out = (ctypes.c_ulonglong * 8)()
        def frame_stats(tag):
            torch.cuda.synchronize(); lib.agx_debug_ray_stats(out, 1)
            for _ in range(5): sen.raycast()
            torch.cuda.synchronize(); lib.agx_debug_ray_stats(out, 1)
            kt = bench.kernel_time_raycast(t)
            print(f"{wl:9s} {obstacles:10s} {tag:12s}: node visits/packet {out[1]/out[0]:6.1f}  leaf tests/packet {out[2]/out[0]:5.1f}  raycast {kt*1e3:.3f} ms (n={n}, stats build)")
        p = _lib.dptr
        def rebuild():
            _lib.check(env._lib.agx_bvh_build(n, sc.num_tris, 12, p(sc.tri_world), None, p(sc.bvh_nodes), p(sc.bvh_work), env._stream()))
        rebuild(); frame_stats("morton")
        tw = sc.tri_world.cpu().numpy().reshape(n, -1, 12, 3, 3)
        lo, hi = tw.min(axis=(2, 3)), tw.max(axis=(2, 3))
        codes = np.stack([sah_codes(lo[e], hi[e]) for e in range(n)])
        parked = lo[..., 0] < -900
        codes = np.where(parked, np.uint32(0xFFFFFFFE), codes >> np.uint32(1)).astype(np.uint32)  # keep bit 31 clear like the product codes
        tc = torch.from_numpy(codes.view(np.int32)).to("cuda:0")
        lib.agx_debug_set_obj_codes(ctypes.c_void_p(tc.data_ptr()))
        rebuild(); frame_stats("host SAH")
        lib.agx_debug_set_obj_codes(ctypes.c_void_p(0))
        del t
