"""Round 6: multi-primitive scenes (cylinders in chunks of 12 triangles + boxes) under the object-level build vs the triangle-level
LBVH: ray-cast frame and refresh times.   python profiles/forest_probe_r06.py [num_envs]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forest import _forest_cfg  # noqa: E402

from aerial_gym_simulator_amd.registry.env_registry import env_config_registry  # noqa: E402
from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
env_config_registry.register("forest_env_fixture", _forest_cfg(num_trees=6))
for box_objects in (True, False):
    import random

    random.seed(2)
    torch.manual_seed(2)
    env = SimBuilder().build_env("base_sim", "forest_env_fixture", "base_quadrotor_with_camera_64x48", "lee_velocity_control", "cuda:0",
                                 args={"bvh_box_objects": box_objects}, num_envs=n)
    env.reset()
    a = torch.zeros(n, 4, device="cuda:0")
    for _ in range(3):
        env.step(actions=a)
        env.post_reward_calculation_step()
    torch.cuda.synchronize()
    sen = env.robot_manager.warp_sensor
    st, sp = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        st.record()
        for _ in range(10):
            sen.raycast(env._stream(), fuse_limits=sen.limits_fusable())
        sp.record()
        torch.cuda.synchronize()
        best = min(best, st.elapsed_time(sp) / 10)
    # a full rebuild of every env's tree
    g = env.global_tensor_dict
    g["reset_mask"].fill_(1)
    g["reset_flag"][env._parity] = 1
    torch.cuda.synchronize()
    st.record()
    env.asset_manager.reset_masked(env)
    sp.record()
    torch.cuda.synchronize()
    print(json.dumps({"num_envs": n, "tris_per_env": env.scene.num_tris, "box_objects_flag": box_objects, "raycast_ms": best,
                      "refresh_all_envs_ms": st.elapsed_time(sp)}), flush=True)
    del env
