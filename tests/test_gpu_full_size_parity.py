"""Parity at the BENCHMARK configurations, every env -- not a sample:

  configs[1]  8192 envs, position-setpoint task, Lee position control: 50 free-running task.step() calls in bench.py's own
              mode (sync-free device RNG, fused fast path) incl. two full reset waves, vs the oracle's env loop: bit-identical
              state / thrust / reward / observation / flags for all 8192 envs at every step;
  configs[2]  8192 envs, navigation task, 100 boxes + 6 walls, 64 x 48 depth + segmentation camera: 3 env steps, teacher-forced
              step by step (tests/test_gpu_nav_task.py run_nav_case): dynamics (10 sub-steps), collision flags, reward, the
              partial (crash) and full (truncation) reset waves -> compacted BVH rebuild, scene transform, the FULL frame of
              every env (25.2 M rays per step vs the oracle's own BVH), post-processing, observation;
  configs[3]  4096 envs, fully-actuated octarotor, 32 x 512 range + segmentation LiDAR (67 M rays per frame), 2 env steps, every
              env its own scene.
Wall time of the last GPU run is printed by each test (pytest -s) and recorded in profiles/r04_full_size_parity.txt."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def npy(t):
    return np.ascontiguousarray(t.detach().cpu().numpy())


def test_config1_every_env_of_8192_for_50_steps(orc):
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry
    from oracle_env import OraclePositionEnv

    t0 = time.time()
    n, T, ep_len, seed = 8192, 50, 20, 0x5EED0123456789
    old = (cfg.device, cfg.controller_name, cfg.episode_len_steps, cfg.args)
    cfg.device, cfg.controller_name, cfg.episode_len_steps = DEV, "lee_position_control", ep_len
    cfg.args = {"strict_rng": False, "rng_seed": seed}  # bench.py's mode: device Philox streams, no host sync
    try:
        task = task_registry.make_task("position_setpoint_task", seed=1, num_envs=n, headless=True)
        task.reset()
        env = task.sim_env
        g = env.global_tensor_dict
        robot = env.robot_manager.robot
        mm = robot.control_allocator.motor_model
        ctrl = robot.controller
        pd = dict(robot.params_dict)
        M = pd["num_motors"]
        gains = [np.tile(((np.array(ctrl.gains_max, np.float32) + np.array(ctrl.gains_min, np.float32)) / np.float32(2))[3 * k:3 * k + 3], (n, 1))
                 for k in range(4)]
        ranges = dict(mm.ranges)
        ranges.setdefault("thrust", (float(pd["min_thrust"]), float(pd["max_thrust"])))
        o = OraclePositionEnv(pd, n, ep_len, gains, robot.min_init_state, robot.max_init_state, ranges)
        # the run starts from the product's state after task.reset() (the reset path itself is what the two waves below check)
        o.state[:], o.thrust[:], o.kT[:] = npy(g["robot_state_tensor"]), npy(mm.current_motor_thrust), npy(mm.motor_thrust_constant)
        o.tau_inc[:], o.tau_dec[:] = npy(mm.motor_time_constants_increasing), npy(mm.motor_time_constants_decreasing)
        o.bmin[:], o.bmax[:] = npy(g["env_bounds_min"]), npy(g["env_bounds_max"])
        o.sim_steps[:] = npy(g["sim_steps"])
        o.euler, o.qveh, o.vveh, o.vbody, o.wbody = orc.update_states(o.state)
        episodes = npy(g["episode_count"]).astype(np.int32)
        e = env.cfg.env
        bcfg = [np.array(x, np.float32) for x in (e.lower_bound_min, e.lower_bound_max, e.upper_bound_min, e.upper_bound_max)]
        agen = torch.Generator(device=DEV).manual_seed(11)
        actions = [torch.rand(n, 4, device=DEV, generator=agen) * 2 - 1 for _ in range(16)]
        waves = 0
        for t in range(T):
            will_reset = (o.sim_steps + 1) > ep_len
            draws = None
            if will_reset.any():
                ub = orc.rng_fill(seed, episodes, orc.RNG_BOUNDS, 6)
                m = will_reset
                o.bmin[m] = ((bcfg[1] - bcfg[0]) * ub[:, :3] + bcfg[0])[m]
                o.bmax[m] = ((bcfg[3] - bcfg[2]) * ub[:, 3:] + bcfg[2])[m]
                mot = orc.rng_fill(seed, episodes, orc.RNG_MOTOR, 4 * M).reshape(n, M, 4)
                draws = (orc.rng_fill(seed, episodes, orc.RNG_STATE, 13), np.ascontiguousarray(mot[..., 0]), np.ascontiguousarray(mot[..., 1]),
                         np.ascontiguousarray(mot[..., 2]), np.ascontiguousarray(mot[..., 3]))
            a = actions[t % 16]
            obs, rew, term, trunc, info = task.step(a)
            o_obs, o_rew, o_crash, o_trunc, o_mask, _ = o.step(npy(a), draws)
            assert np.array_equal(o_mask.astype(bool), will_reset), t
            episodes = episodes + o_mask.astype(np.int32)
            waves += int(o_mask.all())
            for name, got, ref in (("state", g["robot_state_tensor"], o.state), ("thrust", mm.current_motor_thrust, o.thrust),
                                   ("reward", rew, o_rew), ("obs", obs["observations"], o_obs), ("kT", mm.motor_thrust_constant, o.kT),
                                   ("episode_count", g["episode_count"], episodes), ("sim_steps", g["sim_steps"], o.sim_steps)):
                got = npy(got)
                if not np.array_equal(got, ref.astype(got.dtype) if ref.dtype != got.dtype else ref):
                    bad = np.argwhere(got != ref)
                    raise AssertionError(f"step {t}: {name} differs in {len(bad)} of {got.size} entries, first {bad[0]}: "
                                         f"{got[tuple(bad[0])]!r} vs {ref[tuple(bad[0])]!r}")
            assert np.array_equal(npy(term), o_crash.astype(bool)) and np.array_equal(npy(trunc), o_trunc.astype(bool)), t
        assert waves == 2  # every env was reset twice inside the run (steps 21 and 42)
        print(f"\nconfigs[1]: {n} envs x {T} steps, every env bit-identical to the oracle loop; {time.time() - t0:.1f} s wall")
    finally:
        cfg.device, cfg.controller_name, cfg.episode_len_steps, cfg.args = old


def test_config2_every_env_of_8192_full_frames(orc, parity):
    from test_gpu_nav_task import run_nav_case

    t0 = time.time()
    n = 8192
    # episode_len 2: steps 1-2 reset the envs that crashed (a few hundred dirty scenes -> compacted rebuild list), step 3
    # truncates every env (all 8192 scenes rebuilt)
    st = run_nav_case(orc, parity, "config3_camera_8192", "navigation_task", "navigation_task_config", n, 3, episode_len=2,
                      all_obstacles=True, use_bvh=True, min_resets=n)
    assert st["crashes"] >= 50
    print(f"\nconfigs[2]: {n} envs x 3 steps, {3 * n * 64 * 48 / 1e6:.1f} M rays, every pixel of every env bit-identical; "
          f"{st['resets']} resets, {st['crashes']} crashes; {time.time() - t0:.1f} s wall")


def test_config3_every_env_of_4096_lidar_frames(orc, parity):
    from test_gpu_nav_task import run_nav_case

    t0 = time.time()
    n = 4096
    st = run_nav_case(orc, parity, "config4_fully_actuated_lidar_4096", "navigation_task_fully_actuated_lidar",
                      "fully_actuated_lidar_navigation_task_config", n, 2, episode_len=1, all_obstacles=True, use_bvh=True, min_resets=n)
    print(f"\nconfigs[3]: {n} envs (every env its own scene) x 2 steps, {2 * n * 32 * 512 / 1e6:.1f} M rays, every range and id bit-identical; "
          f"{st['resets']} resets, {st['crashes']} crashes; {time.time() - t0:.1f} s wall")
