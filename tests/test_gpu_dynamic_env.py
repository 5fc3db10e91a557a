"""kinematic obstacles (EnvManager.step(actions, env_actions)) -- integration vs the oracle and the
sensors / collision model seeing the moved geometry (examples/dynamic_env_example.py)."""
import numpy as np
import pytest
import torch
from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_assets_integrate_vs_oracle(orc):
    from aerial_gym_simulator_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(4)
    n, K, k, dt = 7, 9, 10, 0.01
    st = np.zeros((n, K, 13), np.float32)
    st[..., 0:3] = rng.uniform(-5, 5, (n, K, 3))
    q = rng.normal(size=(n, K, 4)).astype(np.float32)
    st[..., 3:7] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    tw = rng.uniform(-2, 2, (n, K, 6)).astype(np.float32)
    tw[0, 0, 3:6] = 0.0  # pure translation: orientation must stay bit-identical
    t_st, t_tw = torch.from_numpy(st.copy()).to(DEV), torch.from_numpy(tw).to(DEV)
    _lib.check(lib.agx_assets_integrate(n, K, _lib.dptr(t_st), _lib.dptr(t_tw), dt, k, _lib.current_stream(DEV)))
    torch.cuda.synchronize()
    got = t_st.cpu().numpy()
    ref = orc.assets_integrate(st.copy(), tw, dt, k)
    assert rel_err(got, ref) < 2e-6
    assert np.array_equal(got[..., 7:13], tw) and np.array_equal(got[0, 0, 3:7], st[0, 0, 3:7])
    assert np.allclose(got[..., 0:3], st[..., 0:3] + tw[..., 0:3] * (k * dt), atol=1e-5)
    assert np.allclose(np.linalg.norm(got[..., 3:7], axis=-1), 1.0, atol=1e-6)


def test_dynamic_env_obstacles_move_sensors_and_collisions_follow(orc):
    import random

    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

    random.seed(3)
    torch.manual_seed(3)
    n = 4
    env = SimBuilder().build_env("base_sim", "dynamic_env", "base_quadrotor_with_camera_64x48", "lee_velocity_control", DEV, num_envs=n)
    env.reset()
    g = env.get_obs()
    K = env.scene.num_assets
    assert K == 35 and g["num_env_actions"] == 6
    a = torch.zeros(n, 4, device=DEV)
    twist = torch.zeros(n, K, 6, device=DEV)
    twist[:, :, 0] = -1.0  # dynamic_env_example.py:36: all obstacles drift along -x
    twist[:, :, 5] = 0.5
    st0 = g["env_asset_state_tensor"].clone()
    env.step(actions=a, env_actions=twist)
    env.post_reward_calculation_step()
    st1 = g["env_asset_state_tensor"]
    moved = ~(g["reset_mask"].bool())  # envs that were not reset in this step keep the integrated poses
    k, dt = env.cfg.env.num_physics_steps_per_env_step_mean, g["dt"]
    assert torch.allclose(st1[moved, :, 0], st0[moved, :, 0] - k * dt, atol=1e-5)
    assert torch.equal(st1[moved][:, :, 7:13], twist[moved])
    assert g["env_actions"] is twist and torch.equal(g["prev_env_actions"], twist)
    # geometry followed: triangles == oracle transform of the new poses; depth image == oracle ray-cast of them
    sc, sen = env.scene, env.robot_manager.warp_sensor
    tris = orc.scene_transform(sc.tri_local.cpu().numpy(), sc.tri_asset.cpu().numpy(), st1.cpu().numpy())
    assert np.array_equal(sc.tri_world.cpu().numpy(), tris)
    kinv, cx, cy = orc.camera_kinv(64, 48, sen.cfg.horizontal_fov_deg)
    ref, ref_seg = orc.raycast_camera(64, 48, kinv, sen.cfg.max_range, cx, cy, "depth", sen.sensor_position.cpu().numpy(),
                                      sen.sensor_orientation.cpu().numpy(), tris, sc.tri_seg.cpu().numpy())
    ref = orc.sensor_postprocess(ref, sen.cfg.min_range, sen.cfg.max_range, sen.cfg.far_out_of_range_value, sen.cfg.near_out_of_range_value,
                                 sen.cfg.normalize_range)
    assert np.array_equal(g["segmentation_pixels"].cpu().numpy(), ref_seg) and np.array_equal(g["depth_range_pixels"].cpu().numpy(), ref)
    # an obstacle driven into a hovering robot produces a crash
    robot_p = g["robot_position"].clone()
    st = g["env_asset_state_tensor"]
    st[:, 0, 0:3] = robot_p + torch.tensor([1.0, 0.0, 0.0], device=DEV)  # 1 m in front of the robot
    st[:, 0, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0], device=DEV)
    twist.zero_()
    twist[:, 0, 0] = -4.0  # closes 0.4 m per env step
    crashed = torch.zeros(n, dtype=torch.bool, device=DEV)
    for _ in range(4):
        env.step(actions=a, env_actions=twist)
        crashed |= g["crashes"].clone()
        env.post_reward_calculation_step()
    assert bool(crashed.all())
