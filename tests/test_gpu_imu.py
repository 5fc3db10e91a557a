"""GPU parity of the IMU kernels through the C ABI: vs the reference's golden chain and the
CPU oracle; device-generator statistics; a hover known-answer test through the reference API."""
import numpy as np
import pytest
import torch
from conftest import golden_params, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _args(g, world_frame, gravity=(0.0, 0.0, -9.81)):
    from aerial_gym_simulator_amd import _lib

    A = _lib.AgxImuArgs()
    for i in range(6):
        A.bias_std[i], A.noise_std[i], A.max_value[i], A.max_bias_init[i] = g["bias_std"][i], g["noise_std"][i], g["max_value"][i], g["max_bias_init"][i]
    for i in range(3):
        A.min_rot[i], A.max_rot[i], A.g_world[i] = np.deg2rad(g["min_rot_deg"][i]), np.deg2rad(g["max_rot_deg"][i]), gravity[i]
    A.sqrt_dt, A.mass = float(np.sqrt(0.01)), float(g["mass"])
    A.world_frame, A.enable_noise, A.enable_bias = int(world_frame), 1, 1
    return A


@pytest.mark.parametrize("frame", ["body", "world"])
def test_imu_chain_vs_reference_and_oracle(orc, frame):
    from aerial_gym_simulator_amd import _lib
    from gpu_harness import DynHarness

    g = load_golden("imu_sensor")
    n = g["body_force"].shape[1]
    H = DynHarness(golden_params(load_golden("step_quad_velocity")), n)
    A = _args(g, frame == "world")
    p = _lib.dptr
    bias, sq, meas = torch.zeros(n, 6, device=DEV), torch.zeros(n, 4, device=DEV), torch.zeros(n, 6, device=DEV)
    body_force = torch.zeros(3, n, device=DEV)
    H.B.body_force = p(body_force)
    H.reset_mask.fill_(1)
    H.reset_flag.fill_(1)
    ub, ur = T(g[frame + "_u_bias"]), T(g[frame + "_u_rot"])
    _lib.check(H.lib.agx_imu_reset(H.B, n, A, p(ub), p(ur), p(bias), p(sq), H.stream()))
    torch.cuda.synchronize()
    assert rel_err(bias.cpu().numpy(), g[frame + "_bias0"]) < 1e-6 and rel_err(sq.cpu().numpy(), g[frame + "_sensor_quat"]) < 1e-6
    sq.copy_(T(g[frame + "_sensor_quat"]))  # continue from the reference's exact mount quaternions
    bias.copy_(T(g[frame + "_bias0"]))
    m, gw = float(g["mass"]), np.float32([0, 0, -9.81])
    ob = g[frame + "_bias0"].copy()
    for k in range(g[frame + "_force"].shape[0]):
        q, wb, f = g[frame + "_quat"][k], g[frame + "_wbody"][k], g[frame + "_force"][k]
        state = np.zeros((n, 13), np.float32)
        state[:, 3:7] = q
        derived = np.zeros((n, 16), np.float32)
        derived[:, 13:16] = wb
        H.set(state=state, derived=derived)
        # the force sensor reads f = body_force + m R^T g: feed body_force = f - m R^T g
        state64 = q.astype(np.float64)
        gb = np.stack([orc_rotinv(state64[i], gw.astype(np.float64)) for i in range(n)]).astype(np.float32)
        body_force.copy_(T((f - np.float32(m) * gb).T.copy()))
        zn, zb = T(g[frame + "_z_noise"][k]), T(g[frame + "_z_bias"][k][None])
        _lib.check(H.lib.agx_imu_update(H.B, n, 1, A, p(sq), p(zn), p(zb), p(bias), p(meas), H.stream()))
        torch.cuda.synchronize()
        got = meas.cpu().numpy()
        ref = orc.imu_update(m, gw, np.float32(np.sqrt(0.01)), frame == "world", 1, 1, g["bias_std"], g["noise_std"], g["max_value"], f, q,
                             wb, g[frame + "_sensor_quat"], g[frame + "_z_noise"][k], g[frame + "_z_bias"][k], ob)
        assert rel_err(got, ref) < 1e-5, k                       # oracle
        assert rel_err(got, g[frame + "_meas"][k]) < 1e-5, k     # the reference's own output
    assert rel_err(bias.cpu().numpy(), g[frame + "_bias_end"]) < 1e-6


def orc_rotinv(q, v):
    """quat_rotate_inverse in float64 (test-side input preparation only)"""
    qv, w = q[:3], q[3]
    return v * (2 * w * w - 1) - np.cross(qv, v) * w * 2 + qv * np.dot(qv, v) * 2


def test_imu_device_generator_statistics():
    """sync-free mode: Box-Muller normals from the Philox stream -- white noise with the configured density,
    bias random walk with variance k * (bias_std^2 dt) per env step."""
    from aerial_gym_simulator_amd import _lib
    from gpu_harness import DynHarness

    g = load_golden("imu_sensor")
    n, k = 1 << 16, 10
    H = DynHarness(golden_params(load_golden("step_quad_velocity")), n)
    A = _args(g, False, gravity=(0.0, 0.0, 0.0))
    p = _lib.dptr
    state = np.zeros((n, 13), np.float32)
    state[:, 6] = 1.0
    H.set(state=state)
    bias, sq, meas = torch.zeros(n, 6, device=DEV), torch.zeros(n, 4, device=DEV), torch.zeros(n, 6, device=DEV)
    sq[:, 3] = 1.0
    body_force = torch.zeros(3, n, device=DEV)
    H.B.body_force = p(body_force)
    H.B.rng_seed = 2024
    steps = 20
    acc = []
    for s in range(steps):
        H.B.step_counter = s
        _lib.check(H.lib.agx_imu_update(H.B, n, k, A, p(sq), None, None, p(bias), p(meas), H.stream()))
        acc.append(meas.clone())
    torch.cuda.synchronize()
    b = bias.cpu().numpy().astype(np.float64)
    exp_bias_std = g["bias_std"].astype(np.float64) * 0.1 * np.sqrt(k * steps)
    assert np.all(np.abs(b.std(axis=0) / exp_bias_std - 1.0) < 0.02) and np.all(np.abs(b.mean(axis=0)) < 0.02 * exp_bias_std)
    noise = (acc[-1] - bias).cpu().numpy().astype(np.float64)  # zero motion, zero gravity: measurement = bias + noise
    exp_noise_std = g["noise_std"].astype(np.float64) / 0.1
    assert np.all(np.abs(noise.std(axis=0) / exp_noise_std - 1.0) < 0.02) and np.all(np.abs(noise.mean(axis=0)) < 0.02 * exp_noise_std)
    c = np.corrcoef(noise.T)
    assert np.abs(c - np.eye(6)).max() < 0.02  # channels are independent
    assert np.abs(np.corrcoef(acc[0].cpu().numpy()[:, 0], acc[1].cpu().numpy()[:, 0])[0, 1]) < 0.02  # and white over steps


def test_imu_hover_known_answer():
    """base_quadrotor_with_imu at its position set-point: thrust = m g, so the accelerometer reads +g along body z
    (specific force), the gyro ~0; the `imu_measurement` key of the reference's tensor dict."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

    n = 32
    env = SimBuilder().build_env("base_sim", "empty_env", "base_quadrotor_with_imu", "lee_position_control", DEV, num_envs=n)
    env.reset()
    g = env.get_obs()
    st = g["robot_state_tensor"]
    st[:, 7:13] = 0.0  # at rest
    target = torch.cat([g["robot_position"].clone(), torch.zeros(n, 1, device=DEV)], dim=1)
    for _ in range(200):  # motors spin up, attitude settles
        env.step(actions=target)
    imu = g["imu_measurement"].cpu().numpy()
    assert imu.shape == (n, 6)
    assert np.abs(imu[:, 2] - 9.81).max() < 0.1 and np.abs(imu[:, 0:2]).max() < 0.55  # mount jitter: up to 2 deg about x and y -> 9.81 sin(2.83 deg) = 0.48
    assert np.abs(imu[:, 3:6]).max() < 0.1
