"""Experiment (round 3): how much of the ray-cast's time is tree quality?  The device builds an LBVH per env (Morton order of
the objects' centres, csrc/agx_scene.hip); this probe rebuilds the SAME leaves (the two-triangle leaves the device emitted,
with their boxes) into a top-down full-sweep SAH tree on the host, writes it in the device's 64-byte node format, and times
the unchanged k_raycast on both.  The ray-cast result is defined over all triangles (closest hit, face-index tie-break), so
the images must be bit-identical whatever the tree.

    python profiles/sah_probe.py [num_envs] > gpurun_out/sah_probe.json
"""
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def leaves_of(nodes):
    """(lo [P,3], hi [P,3], tri [P], second [P]) of the leaves reachable from node 0"""
    ni = nodes.view(np.int32)
    lo, hi, tri, sec, visits = [], [], [], [], 0
    stack = [0]
    while stack:
        i = stack.pop()
        visits += 1
        for (l0, h0, cref, sref) in ((0, 4, 3, 11), (8, 12, 7, 15)):
            c = int(ni[i, cref])
            if c >= 0:
                stack.append(c)
            else:
                lo.append(nodes[i, l0:l0 + 3].copy()); hi.append(nodes[i, h0:h0 + 3].copy())
                tri.append(c); sec.append(int(ni[i, sref]))
    return np.array(lo), np.array(hi), np.array(tri, np.int32), np.array(sec, np.int32), visits


def area(lo, hi):
    d = np.maximum(hi - lo, 0.0)
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]


def sah_build(args):
    nodes, = args
    lo, hi, tri, sec, _ = leaves_of(nodes)
    P = len(tri)
    out = np.zeros_like(nodes)
    oi = out.view(np.int32)
    cen = 0.5 * (lo + hi)
    counter = [0]
    depth_max = [0]

    def emit(idx, depth):
        """idx: the primitives of this subtree (>= 2); returns the node index"""
        me = counter[0]
        counter[0] += 1
        depth_max[0] = max(depth_max[0], depth)
        best = None
        n = len(idx)
        for ax in range(3):
            order = idx[np.argsort(cen[idx, ax], kind="stable")]
            l_lo = np.minimum.accumulate(lo[order], 0); l_hi = np.maximum.accumulate(hi[order], 0)
            r_lo = np.minimum.accumulate(lo[order][::-1], 0)[::-1]; r_hi = np.maximum.accumulate(hi[order][::-1], 0)[::-1]
            k = np.arange(1, n)
            cost = area(l_lo[:-1], l_hi[:-1]) * k + area(r_lo[1:], r_hi[1:]) * (n - k)
            j = int(np.argmin(cost))
            if best is None or cost[j] < best[0]:
                best = (float(cost[j]), order[:j + 1], order[j + 1:])
        for side, sub in enumerate(best[1:]):
            l0, h0, cref, sref = ((0, 4, 3, 11), (8, 12, 7, 15))[side]
            out[me, l0:l0 + 3] = lo[sub].min(0)
            out[me, h0:h0 + 3] = hi[sub].max(0)
            if len(sub) == 1:
                oi[me, cref] = tri[sub[0]]
                oi[me, sref] = sec[sub[0]]
            else:
                oi[me, cref] = emit(sub, depth + 1)
                oi[me, sref] = -1
        return me

    sys.setrecursionlimit(10000)
    emit(np.arange(P), 1)
    return out, P, depth_max[0]


def main():
    pool = Pool(min(os.cpu_count() or 8, 128))  # forked before the process touches the GPU
    import torch

    import bench

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    task = bench.make_task("depth", n, "cuda:0", False, 0)
    task.reset()
    a = torch.zeros(n, 4, device="cuda:0")
    for _ in range(3):
        task.step(a)
    env = task.sim_env
    sc = env.scene
    sensor = env.robot_manager.warp_sensor
    torch.cuda.synchronize()
    t_lbvh = bench.kernel_time_raycast(task)
    img0 = sensor.pixels.clone()
    nodes0 = sc.bvh_nodes.clone()
    host = nodes0.cpu().numpy()
    res = pool.map(sah_build, [(host[e],) for e in range(n)], chunksize=4)
    pool.close()
    new = np.stack([r[0] for r in res])
    sc.bvh_nodes.copy_(torch.from_numpy(new).to(sc.bvh_nodes.device))
    torch.cuda.synchronize()
    t_sah = bench.kernel_time_raycast(task)
    img1 = sensor.pixels.clone()
    same = bool(torch.equal(img0.view(torch.int32), img1.view(torch.int32)))
    sc.bvh_nodes.copy_(nodes0)
    torch.cuda.synchronize()
    t_lbvh2 = bench.kernel_time_raycast(task)
    print(json.dumps({"num_envs": n, "leaves_per_env": int(np.mean([r[1] for r in res])), "sah_depth_max": int(max(r[2] for r in res)),
                      "raycast_us_lbvh": t_lbvh * 1e6, "raycast_us_lbvh_again": t_lbvh2 * 1e6, "raycast_us_sah": t_sah * 1e6,
                      "sah_over_lbvh": t_sah / min(t_lbvh, t_lbvh2), "images_bit_identical": same}))


if __name__ == "__main__":
    main()
