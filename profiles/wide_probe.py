"""Experiment (round 3): 4-wide BVH nodes for the packet traversal.  The device's binary LBVH is converted on the host (every
record takes its grandchildren: 2 .. 4 entries, a leaf child keeps its place), the experimental build of the library
(-DAGX_RAY_WIDE=1, csrc/agx_raycast.hip) traverses the wide records, and the frame is compared bit for bit with the frame of
the default library on the same scene and poses.

    python profiles/wide_probe.py base  depth 2048 gpurun_out/wide_base.npz      # default library: time + reference frame
    AGX_LIB_PATH=aerial_gym_simulator_amd/lib/libaerialgym_hip_wide.so \\
    python profiles/wide_probe.py wide  depth 2048 gpurun_out/wide_base.npz      # experimental library on converted nodes
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NO_ENTRY = np.int32(-2 ** 31)


def to_wide(nodes):
    """[N, M, 16] binary records -> [N, M, 32] wide records (same index space; unreachable records stay unreachable)"""
    ni = nodes.view(np.int32)
    N, M, _ = nodes.shape
    wide = np.zeros((N, M, 32), np.float32)
    wi = wide.view(np.int32)
    wi[..., 24:28] = NO_ENTRY
    wi[..., 28:32] = -1
    for side, (l0, h0, cref, sref) in enumerate(((0, 4, 3, 11), (8, 12, 7, 15))):
        c = ni[..., cref]
        internal = (c >= 0)[..., None]
        child = np.take_along_axis(nodes, np.where(c >= 0, c, 0)[..., None], axis=1)
        ci = child.view(np.int32)
        k = 2 * side
        lo, hi = nodes[..., l0:l0 + 3], nodes[..., h0:h0 + 3]
        # first entry of the pair: the child's left box, or the leaf child itself
        wide[..., 6 * k:6 * k + 3] = np.where(internal, child[..., 0:3], lo)
        wide[..., 6 * k + 3:6 * k + 6] = np.where(internal, child[..., 4:7], hi)
        wi[..., 24 + k] = np.where(internal[..., 0], ci[..., 3], c)
        wi[..., 28 + k] = np.where(internal[..., 0], ci[..., 11], ni[..., sref])
        # second entry: the child's right box, or nothing
        wide[..., 6 * k + 6:6 * k + 9] = np.where(internal, child[..., 8:11], lo)
        wide[..., 6 * k + 9:6 * k + 12] = np.where(internal, child[..., 12:15], hi)
        wi[..., 25 + k] = np.where(internal[..., 0], ci[..., 7], NO_ENTRY)
        wi[..., 29 + k] = np.where(internal[..., 0], ci[..., 15], -1)
    return wide


def main():
    import torch

    import bench

    which, workload, n, path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    from aerial_gym_simulator_amd.sensors import hip_sensor

    cls = next(v for v in vars(hip_sensor).values() if isinstance(v, type) and "raycast" in vars(v))
    real_raycast, hold = cls.raycast, [which == "wide"]  # the experimental library must not traverse binary records

    def raycast(self, *a, **k):
        return None if hold[0] else real_raycast(self, *a, **k)

    cls.raycast = raycast
    task = bench.make_task(workload, n, "cuda:0", False, 0)
    task.reset()
    a = torch.zeros(n, task.task_config.action_space_dim, device="cuda:0")
    for _ in range(3):
        task.step(a)
    env = task.sim_env
    sc = env.scene
    sensor = env.robot_manager.warp_sensor
    torch.cuda.synchronize()
    out = {"which": which, "workload": workload, "num_envs": n, "library": os.environ.get("AGX_LIB_PATH", "default")}
    if which == "wide":
        wide = to_wide(sc.bvh_nodes.cpu().numpy())
        sc.bvh_nodes = torch.from_numpy(wide).to("cuda:0")
        torch.cuda.synchronize()
        hold[0] = False
    t = bench.kernel_time_raycast(task)
    torch.cuda.synchronize()
    px = sensor.pixels.cpu().numpy()
    seg = sensor.segmentation_pixels.cpu().numpy() if getattr(sensor, "segmentation_pixels", None) is not None else np.zeros(1, np.int32)
    out["raycast_us"] = t * 1e6
    nodes = sc.bvh_nodes.cpu().numpy() if which != "wide" else np.zeros(1, np.float32)
    if which == "base":
        np.savez(path, px=px, seg=seg, nodes=nodes)
    else:
        ref = np.load(path)
        out["frame_bit_identical"] = bool(np.array_equal(ref["px"].view(np.int32), px.view(np.int32)) and np.array_equal(ref["seg"], seg))
        out["pixels_differing"] = int((ref["px"].view(np.int32) != px.view(np.int32)).sum())
        if which == "variant" and "nodes" in ref.files:  # same builder input -> are the TREES the same?
            out["bvh_nodes_bit_identical"] = bool(np.array_equal(ref["nodes"].view(np.int32), nodes.view(np.int32)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
