"""GPU parity of the LiDAR-navigation kernels through the C ABI: vs the CPU oracle and
vs the golden vectors generated from the reference's own code, plus the task end to end."""
import ctypes as C

import numpy as np
import pytest
import torch
from conftest import golden_params, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RNG_LIDAR_NOISE, RNG_OBS_NOISE = 5, 6


def T(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dt is None else t.to(dt)


def _harness(n):
    from gpu_harness import DynHarness

    return DynHarness(golden_params(load_golden("step_quad_velocity")), n)


def test_reward_lidar_navigation_vs_reference_and_oracle(orc):
    from aerial_gym_simulator_amd import _lib

    g = load_golden("reward_lidar_navigation")
    n = g["pos_err"].shape[0]
    H = _harness(n)
    # identity vehicle frame at the origin: pos_err = target exactly; euler z = 0: yaw error = ssa(target_yaw)
    state = np.zeros((n, 13), np.float32)
    state[:, 6] = 1.0
    derived = np.zeros((n, 16), np.float32)
    derived[:, 6] = 1.0
    derived[:, 7:10], derived[:, 13:16] = g["vveh"], g["wbody"]
    H.set(state=state, derived=derived)
    H.crashes.copy_(torch.from_numpy(g["crashes"]))
    H.sim_steps.copy_(torch.arange(n, dtype=torch.int32) % 130)
    tgt, tyaw = T(g["pos_err"].T.copy()), T(g["yaw_error"])
    act, pact, ttc = T(g["action"]), T(g["prev_action"]), T(g["time_to_collision"])
    pe, ppe = T(np.full((3, n), 7.0, np.float32)), torch.zeros(3, n, device=DEV)
    rew = torch.zeros(n, device=DEV)
    rp = (C.c_float * 22)(*[float(x) for x in g["rp"]])
    p = _lib.dptr
    _lib.check(H.lib.agx_reward_lidar_navigation(H.B, n, p(tgt), p(tyaw), p(act), p(pact), p(ttc), rp, float(g["curriculum_progress"]),
                                                 p(pe), p(ppe), 110, 1, p(rew), H.stream()))
    torch.cuda.synchronize()
    r = rew.cpu().numpy()
    assert rel_err(r, g["reward"]) < 1e-5  # vs the reference's own compute_reward
    ref = orc.reward_lidar_navigation(g["pos_err"], g["vveh"], g["wbody"], g["yaw_error"], g["crashes"], g["action"], g["prev_action"],
                                      g["time_to_collision"], float(g["curriculum_progress"]), g["rp"])
    assert rel_err(r, ref) < 1e-5
    assert np.array_equal(pe.cpu().numpy().T, g["pos_err"]) and np.all(ppe.cpu().numpy() == 7.0)  # prev <- cur, cur <- new
    trunc = (np.arange(n) % 130) > 110
    assert np.array_equal(H.trunc.cpu().numpy(), trunc)  # flags: bit-exact
    assert np.array_equal(H.reset_mask.cpu().numpy().astype(bool), trunc | g["crashes"])
    assert int(H.reset_flag.cpu()[0]) == 1


def test_lidar_image_obs_bit_exact(orc):
    from aerial_gym_simulator_amd import _lib

    g = load_golden("lidar_image_obs")
    pc = np.ascontiguousarray(g["pointcloud"][:, 0])
    n = pc.shape[0]
    H = _harness(n)
    state = np.zeros((n, 13), np.float32)
    state[:, 0:3], state[:, 6], state[:, 7:10] = g["robot_position"], 1.0, g["robot_linvel"]
    H.set(state=state)
    tpc = T(pc)
    ttc, ds = torch.zeros(n, device=DEV), torch.zeros(n, 320, device=DEV)
    p = _lib.dptr

    def run(noise, device_noise=0):
        _lib.check(H.lib.agx_lidar_image_obs(H.B, n, 48, 120, 3, 6, 10, p(tpc), *[p(t) if t is not None else None for t in noise],
                                             device_noise, p(ttc), p(ds), H.stream()))
        torch.cuda.synchronize()
        return ttc.cpu().numpy(), ds.cpu().numpy()

    got_ttc, got_ds = run([None] * 5)
    ref_ttc, ref_ds = orc.lidar_image_obs(pc, g["robot_position"], g["robot_linvel"])
    assert np.array_equal(got_ttc, ref_ttc) and np.array_equal(got_ds, ref_ds)     # vs oracle: bit-exact
    assert rel_err(got_ttc, g["clean_ttc"]) < 2e-6 and rel_err(got_ds, g["clean_ds"]) < 2e-6  # vs reference
    # the kernel reads a 16-byte-aligned cloud with 16-byte loads (four points per three loads); a cloud that starts 4 bytes off
    # takes the 4-byte path: same bits
    assert tpc.data_ptr() % 16 == 0
    shifted = torch.zeros(tpc.numel() + 1, device=DEV)[1:].view_as(tpc)
    shifted.copy_(tpc)
    assert shifted.data_ptr() % 16 == 4
    aligned = tpc
    tpc = shifted
    off_ttc, off_ds = run([None] * 5)
    tpc = aligned
    assert np.array_equal(off_ttc, got_ttc) and np.array_equal(off_ds, got_ds)
    low = lambda a: np.concatenate([np.zeros((n, 10, 20), np.float32), a], axis=1)  # noqa: E731
    noise_np = [g["noise_mask"], g["noise_val"], g["max_mask"], low(g["low_mask"]), low(g["low_val"])]
    noise_t = [T(a) for a in noise_np]
    got_ttc2, got_ds2 = run(noise_t)
    assert rel_err(got_ds2, g["noisy_ds"]) < 2e-6 and np.array_equal(got_ttc2, got_ttc)
    # device generator: Philox stream RNG_LIDAR_NOISE of (env, step), blocks 2c / 2c+1 of pooled cell c
    H.B.rng_seed, H.B.step_counter = 1234567, 41
    got_ttc3, got_ds3 = run([None] * 5, device_noise=1)
    u = orc.rng_fill(1234567, np.full(n, 41), RNG_LIDAR_NOISE, 8 * 320).reshape(n, 320, 8)
    f32 = np.float32
    nm = (u[..., 0] < f32(0.03)).astype(f32)
    nv = ((f32(10.0) - f32(0.2)) * u[..., 1] + f32(0.2)).astype(f32)
    mm = (u[..., 2] < f32(0.02)).astype(f32)
    lm = (u[..., 3] < f32(0.02)).astype(f32)
    lv = ((f32(1.0) - f32(0.2)) * u[..., 4] + f32(0.2)).astype(f32)
    _, ref_ds3 = orc.lidar_image_obs(pc, g["robot_position"], g["robot_linvel"], noise_mask=nm, noise_val=nv, max_mask=mm,
                                     low_mask=lm, low_val=lv)
    assert np.array_equal(got_ds3, ref_ds3) and np.array_equal(got_ttc3, got_ttc)
    changed = (got_ds3 != got_ds).mean()
    assert 0.03 < changed < 0.09  # ~3 % + 2 % + 2 % of 3/8 of the cells


def test_obs_lidar_navigation_vs_reference_and_oracle(orc):
    from aerial_gym_simulator_amd import _lib

    g = load_golden("obs_lidar_navigation")
    n = g["state"].shape[0]
    H = _harness(n)
    derived = np.concatenate([g["euler"], g["qveh"], np.zeros((n, 3), np.float32), g["vbody"], g["wbody"]], axis=1)
    H.set(state=g["state"], derived=derived, actions=g["actions"])
    tgt, tyaw, uv, ue, ds = T(g["target"].T.copy()), T(g["target_yaw"]), T(g["u_vec"]), T(g["u_euler"]), T(g["downsampled"])
    obs = torch.zeros(n, 337, device=DEV)
    p = _lib.dptr
    _lib.check(H.lib.agx_obs_lidar_navigation(H.B, n, p(tgt), p(tyaw), p(uv), p(ue), p(ds), 320, p(obs), H.stream()))
    torch.cuda.synchronize()
    o = obs.cpu().numpy()
    ref = orc.obs_lidar_navigation(g["state"], g["euler"], g["qveh"], g["vbody"], g["wbody"], g["actions"], g["target"],
                                   g["target_yaw"], g["u_vec"], g["u_euler"], g["downsampled"])
    keep = np.r_[0:6, 7:337]
    for other, tol in ((ref, 1e-6), (g["obs"], 3e-6)):
        d = np.abs(o[:, 6] - other[:, 6])
        assert np.minimum(d, 2 * np.pi - d).max() < 3e-6
        assert rel_err(o[:, keep], other[:, keep]) < tol
    assert np.array_equal(o[:, 17:], g["downsampled"])
    # device generator for the two rand_like draws
    H.B.rng_seed, H.B.step_counter = 99, 7
    _lib.check(H.lib.agx_obs_lidar_navigation(H.B, n, p(tgt), p(tyaw), None, None, p(ds), 320, p(obs), H.stream()))
    torch.cuda.synchronize()
    u6 = orc.rng_fill(99, np.full(n, 7), RNG_OBS_NOISE, 6)
    ref2 = orc.obs_lidar_navigation(g["state"], g["euler"], g["qveh"], g["vbody"], g["wbody"], g["actions"], g["target"],
                                    g["target_yaw"], u6[:, 0:3], u6[:, 3:6], g["downsampled"])
    assert rel_err(obs.cpu().numpy()[:, keep], ref2[:, keep]) < 1e-6


def test_magpie_substeps_vs_oracle_and_reference(orc):
    """magpie + magpie_acceleration_control (root-link wrench mode, randomised gains): fused sub-steps."""
    from gpu_harness import DynHarness

    g = load_golden("step_magpie_acceleration")
    pd = golden_params(g)
    n = g["state"].shape[1]
    P = orc.make_params(pd)
    H = DynHarness(pd, n)
    H.set(kT=g["kT"], tau_inc=g["tau_inc"], tau_dec=g["tau_dec"], state=g["state"][0], thrust=g["thrust_in"][0])
    H.set_gains(g["Kp"], g["Kv"], g["KR"], g["Kw"])
    H.substeps(g["action"][0], 1)
    assert rel_err(H.get("thrust"), g["thrust_out"][0]) < 1e-5   # the reference's own motor-model output
    st, th = g["state"][0].copy(), g["thrust_in"][0].copy()
    orc.substep(P, st, g["action"][0], th, g["kT"], g["tau_inc"], g["tau_dec"], g["Kp"], g["Kv"], g["KR"], g["Kw"],
                disturb_max=g["disturb_max"])
    assert rel_err(H.get("state"), st) < 1e-5
    for _ in range(9):
        orc.substep(P, st, g["action"][0], th, g["kT"], g["tau_inc"], g["tau_dec"], g["Kp"], g["Kv"], g["KR"], g["Kw"],
                    disturb_max=g["disturb_max"])
    H.substeps(g["action"][0], 9)
    assert rel_err(H.get("state"), st) < 1e-4 and rel_err(H.get("thrust"), th) < 1e-4


@pytest.mark.parametrize("strict", [False, True])
def test_lidar_navigation_task_runs(strict):
    """Reference names end to end: task_registry.make_task("lidar_navigation_task") = magpie +
    magpie_acceleration_control + RS-LiDAR dome in env_with_lidar_nav_obstacles; 337-D observation."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import lidar_navigation_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    cfg.device, cfg.args = DEV, {"strict_rng": strict}
    try:
        n = 48
        task = task_registry.make_task("lidar_navigation_task", seed=3, num_envs=n, headless=True)
        obs, *_ = task.reset()
        assert obs["observations"].shape == (n, 337) and task.task_config.robot_name == "magpie"
        a = torch.rand(n, 4, device=DEV) * 2 - 1
        resets = 0
        for i in range(140):
            obs, rew, term, trunc, info = task.step(a)
            resets += int((term | trunc).sum())
        o = obs["observations"]
        assert torch.isfinite(o).all() and torch.isfinite(rew).all()
        assert resets >= n  # episode_len 110: every env ended at least once
        inv = o[:, 17:]
        assert float(inv.min()) >= 1.0 / 20.0 - 1e-6 and float(inv.max()) <= 5.0 + 1e-5  # 1 / [0.2, 10 (+10 noise)]
        assert float(task.time_to_collision.min()) >= 0.0 and float(task.time_to_collision.max()) <= 10.0
        assert torch.equal(task.prev_action, task.action_transformation_function(a))  # the last two actions were identical
        px = task.obs_dict["depth_range_pixels"]
        assert px.shape == (n, 1, 48, 120, 3)
    finally:
        cfg.args = {}
