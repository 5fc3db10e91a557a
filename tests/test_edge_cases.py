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
    assert lib.agx_raycast_camera(1, 1, 8, 8, (C.c_float * 4)(), 10.0, 4, 4, 9, None, None, None, None, None, 12, None, None, None) == -1
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
