"""Edge cases: argument validation of the C ABI (CPU, no compute) and odd sizes on the GPU."""
import ctypes as C

import numpy as np
import pytest
import torch
from conftest import golden_params, load_golden, rel_err


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    from aerial_gym_simulator_amd import _lib

    lib = _lib.load()
    P, B = _lib.AgxRobotParams(), _lib.AgxEnvBuffers()
    P.num_motors, P.num_actions, P.controller = 4, 4, 1
    assert lib.agx_env_step(None, B, 8, None, 1, None, None) < 0                      # null params / actions
    assert lib.agx_env_step(P, B, 0, C.c_void_p(16), 1, None, None) == -1             # num_envs must be > 0
    assert b"num_envs" in lib.agx_last_error()
    assert lib.agx_env_step(P, B, 8, C.c_void_p(16), 99, None, None) == -1            # k out of range
    assert b"k_substeps" in lib.agx_last_error()
    P.num_motors = 5
    B.state = B.derived = B.actions = B.prev_actions = B.motor_thrust = B.crashes = B.truncations = B.sim_steps = 16
    P.num_actions = 4
    assert lib.agx_env_step(P, B, 8, C.c_void_p(16), 1, None, None) == -3             # unsupported motor count
    P.num_motors, P.controller, P.num_actions = 4, 7, 4
    assert lib.agx_env_step(P, B, 8, C.c_void_p(16), 1, None, None) == -1             # fully actuated needs 7 actions
    assert lib.agx_bvh_build(4, 5000, 0, C.c_void_p(16), None, C.c_void_p(16), None, None) == -1  # LDS-resident build limit
    assert b"num_tris" in lib.agx_last_error()
    assert lib.agx_bvh_build(4, 24, 0, C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), None, None) == -1  # masked rebuild needs the work list
    assert lib.agx_raycast_camera(1, 1, 8, 8, (C.c_float * 4)(), 10.0, 4, 4, 9, None, None, None, None, None, 12, None, None, None, None) == -1
    assert lib.agx_bvh_nodes_bytes(3, 1272) == 3 * 1271 * 64


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 65, 1000])
def test_odd_env_counts_and_zero_substeps(orc, n):
    from gpu_harness import DynHarness

    g = load_golden("step_quad_velocity")
    pd = golden_params(g)
    idx = np.arange(n) % g["state"].shape[1]
    P = orc.make_params(pd)
    H = DynHarness(pd, n)
    arr = {k: g[k][idx] for k in ("kT", "tau_inc", "tau_dec", "Kp", "Kv", "KR", "Kw")}
    H.set(kT=arr["kT"], tau_inc=arr["tau_inc"], tau_dec=arr["tau_dec"], state=g["state"][0][idx], thrust=g["thrust_in"][0][idx])
    H.set_gains(arr["Kp"], arr["Kv"], arr["KR"], arr["Kw"])
    act = g["action"][0][idx]
    H.substeps(act, 0)  # k = 0: nothing moves, the step counter still advances (env_manager.py:417-429)
    assert np.array_equal(H.get("state"), g["state"][0][idx]) and int(H.sim_steps.cpu()[0]) == 1
    st, th = g["state"][0][idx].copy(), g["thrust_in"][0][idx].copy()
    for _ in range(32):
        orc.substep(P, st, act, th, arr["kT"], arr["tau_inc"], arr["tau_dec"], arr["Kp"], arr["Kv"], arr["KR"], arr["Kw"], disturb_max=g["disturb_max"])
    H.substeps(act, 32)  # AGX_MAX_SUBSTEPS
    assert rel_err(H.get("state"), st) < 3e-4
    assert np.isfinite(H.get("state")).all()


@pytest.mark.gpu
def test_tiny_and_empty_scenes(orc):
    """T = 2 triangles (smallest tree), rays that hit nothing, far plane in front of the geometry."""
    from test_gpu_raycast import Scene

    a, b, c, d = [3, -1, -1], [3, 1, -1], [3, 1, 1], [3, -1, 1]
    tri_local = np.array([[a + b + c, a + c + d]], np.float32)
    st = np.zeros((1, 1, 13), np.float32)
    st[..., 6] = 1.0
    sc = dict(tri_local=tri_local, tri_asset=np.zeros(2, np.int32), tri_seg=np.array([[5, 6]], np.int32), asset_state=st,
              half=np.ones((1, 1, 3), np.float32))
    S = Scene(sc)
    S.build()
    world = orc.scene_transform(tri_local, sc["tri_asset"], st)
    assert np.array_equal(S.tri_world.cpu().numpy(), world)
    kinv, cx, cy = orc.camera_kinv(16, 12, 60.0)
    frame = orc.quat_from_euler(np.deg2rad(np.array([[-90.0, 0.0, -90.0]], np.float32))).reshape(1, 1, 4)
    pos = np.zeros((1, 1, 3), np.float32)
    for far, any_hit in ((10.0, True), (2.0, False)):  # 2.0: the plate is beyond the far plane -> all misses
        ref = orc.raycast_camera(16, 12, kinv, far, cx, cy, "depth", pos, frame, world, sc["tri_seg"])
        got = S.camera(16, 12, kinv, far, cx, cy, 1, pos, frame)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        assert bool((got[1] >= 0).any()) == any_hit
    yaw_pi = np.array([[0.0, 0.0, 1.0, 0.0]], np.float32)  # looking away from the plate
    quat = orc.quat_mul(yaw_pi, frame.reshape(1, 4)).reshape(1, 1, 4)
    ref = orc.raycast_camera(16, 12, kinv, 10.0, cx, cy, "depth", pos, quat, world, sc["tri_seg"])
    got = S.camera(16, 12, kinv, 10.0, cx, cy, 1, pos, quat)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and not (got[1] >= 0).any()


@pytest.mark.gpu
def test_standalone_env_manager_resets_crashed_and_truncated_envs():
    """The reference's own examples drive EnvManager without a task: `env.step(actions);
    env.reset_terminated_and_truncated_envs()` (examples/benchmark.py), and its tasks set `truncations[:]` in torch
    before `post_reward_calculation_step` (position_setpoint_task.py:172-176).  The reset set is derived right there
    from crashes * reset_on_collision + truncations (env_manager.py:364-371), without any task kernel having run."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

    n, dev = 512, "cuda:0"
    env = SimBuilder().build_env(sim_name="base_sim", env_name="env_with_random_boxes", robot_name="base_quadrotor",
                                 controller_name="lee_velocity_control", device=dev, args={}, num_envs=n, headless=True, use_warp=False)
    env.reset()
    g = env.global_tensor_dict
    a = torch.zeros(n, 4, device=dev)
    a[:, 0] = 3.0  # fly forward into the obstacle field
    crashed_total = reset_total = 0
    for t in range(60):
        env.step(a)
        g["truncations"][:] = env.sim_steps > 25  # a user task's time-out rule, in torch
        want = (g["crashes"] & bool(env.cfg.env.reset_on_collision)) | g["truncations"]
        before = g["episode_count"].clone()
        ids = env.reset_terminated_and_truncated_envs()
        torch.cuda.synchronize()
        assert torch.equal(g["reset_mask"].bool(), want), t
        assert torch.equal(g["episode_count"] - before, want.int()), t  # exactly those envs went through the reset
        assert torch.all(env.sim_steps[want] == 0)
        assert len(ids) == int(want.sum())
        crashed_total += int(g["crashes"].sum())
        reset_total += int(want.sum())
    assert crashed_total > 0 and reset_total > n  # collisions and time-outs both happened


@pytest.mark.gpu
def test_device_rng_and_scenes_do_not_depend_on_the_sharding():
    """10 envs on 3 ranks = shards of 4, 3, 3 (sharding.shard_range gives the remainder to the first ranks): with
    `env_offset` = the shard's first global env, scene seeds, segmentation ids and the device generator's streams are
    those of the unsharded run -- reset states, obstacle layouts and ids concatenate bit for bit."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.sharding import shard_range
    from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

    dev, total, world = "cuda:0", 10, 3

    def build(n, offset, rank):
        e = SimBuilder().build_env(sim_name="base_sim", env_name="env_with_random_boxes", robot_name="base_quadrotor_with_camera_64x48",
                                   controller_name="lee_velocity_control", device=dev, num_envs=n, headless=True, use_warp=True,
                                   args={"rng_seed": 77, "shard_rank": rank, "env_offset": offset})
        e.reset()
        for _ in range(3):
            e.step(torch.zeros(n, 4, device=dev))
            e.global_tensor_dict["truncations"][:] = True  # every env goes through a second, third ... reset
            e.post_reward_calculation_step()
        torch.cuda.synchronize()
        return e

    full = build(total, 0, 0)
    keys = ("robot_state_tensor", "env_bounds_min", "env_bounds_max", "segmentation_pixels", "depth_range_pixels")
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(build(hi - lo, lo, r))
    assert [p.num_envs for p in parts] == [4, 3, 3]
    for k in keys:
        cat = torch.cat([p.global_tensor_dict[k] for p in parts], dim=0)
        assert torch.equal(cat, full.global_tensor_dict[k]), k
    assert torch.equal(torch.cat([p.scene.asset_state for p in parts], dim=0), full.scene.asset_state)
    assert torch.equal(torch.cat([p.scene.tri_seg for p in parts], dim=0), full.scene.tri_seg)
    seg = full.global_tensor_dict["segmentation_pixels"]
    assert len(torch.unique(full.scene.tri_seg[full.scene.tri_seg >= 100])) > total  # distinct ids across envs


@pytest.mark.gpu
def test_reset_assets_launches_beyond_65535_envs():
    """agx_reset_assets puts the env on grid.x: no 65535 cap (HIP's limit for grid.y) on the env count."""
    from aerial_gym_simulator_amd import _lib
    from aerial_gym_simulator_amd._lib import AgxEnvBuffers, AgxResetArgs

    lib = _lib.load()
    n, K, dev = 70000, 3, "cuda:0"
    B, R = AgxEnvBuffers(), AgxResetArgs()
    mask = torch.ones(n, dtype=torch.uint8, device=dev)
    flag = torch.tensor([1, 0], dtype=torch.int32, device=dev)
    ep = torch.zeros(n, dtype=torch.int32, device=dev)
    B.reset_mask, B.reset_flag, B.flag_parity, B.episode_count = _lib.dptr(mask), _lib.dptr(flag), 0, _lib.dptr(ep)
    for i in range(3):
        R.lower_bound_min[i] = R.lower_bound_max[i] = -5.0
        R.upper_bound_min[i] = R.upper_bound_max[i] = 5.0
    R.seed = 5
    lo = torch.zeros(n, K, 13, device=dev)
    hi = torch.zeros(n, K, 13, device=dev)
    lo[..., :3], hi[..., :3] = 0.2, 0.8
    st = torch.zeros(n, K, 13, device=dev)
    _lib.check(lib.agx_reset_assets(B, n, K, R, None, None, None, _lib.dptr(lo), _lib.dptr(hi), K, 0, _lib.dptr(st),
                                    _lib.current_stream(dev)), "agx_reset_assets")
    torch.cuda.synchronize()
    p = st[..., :3]
    placed = p[..., 0] > -999.0  # ~15 % of the envs keep half of their obstacles, the rest is parked at -1000 m
    assert 0.8 < float(placed.float().mean()) < 1.0
    assert float(p[placed].min()) >= -3.0 - 1e-5 and float(p[placed].max()) <= 3.0 + 1e-5
    assert float(p[-1].abs().sum()) > 0 and float(p[0].abs().sum()) > 0  # the last env was reached
    assert len(torch.unique(p[:, 0, 0])) > n // 2  # per-env draws
