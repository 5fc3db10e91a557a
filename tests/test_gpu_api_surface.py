"""GPU: the small API surface closed in round 4.
  * the reference's DEFAULT navigation recipe by name: robot `lmf2` + `lmf2_velocity_control` (navigation_task_config.py:9-10)
  * `base_quad_root_link_control` (robots/__init__.py:43)
  * EnvManager.compute_observations() as its own launch (env_manager.py:358-362) for callers that drive simulate() themselves
  * BaseLeeController.randomize_params(env_ids) outside a reset (base_lee_controller.py:101-118)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_reference_default_navigation_recipe_builds_by_name_and_steps():
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import navigation_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    old = (cfg.robot_name, cfg.controller_name, cfg.device, cfg.args)
    cfg.robot_name, cfg.controller_name, cfg.device, cfg.args = "lmf2", "lmf2_velocity_control", DEV, {}
    try:
        n = 24
        task = task_registry.make_task("navigation_task", seed=2, num_envs=n, headless=True)
        env = task.sim_env
        robot = env.robot_manager.robot
        assert robot.params_dict["root_link_mode"] == 1 and abs(robot.params_dict["mass"] - 1.24) < 1e-6
        sensor = env.robot_manager.warp_sensor
        assert (sensor.cfg.height, sensor.cfg.width) == (135, 240)
        ctrl = robot.controller
        assert ctrl.cfg.randomize_params and getattr(ctrl, "_per_env_gains_bound", False)
        task.reset()
        g = env.global_tensor_dict
        z0 = g["robot_position"][:, 2].clone()
        ever_reset = torch.zeros(n, dtype=torch.bool, device=DEV)
        for _ in range(8):
            obs, rew, term, trunc, info = task.step(torch.zeros(n, 4, device=DEV))  # speed 1 m/s forward, level, no yaw rate
            ever_reset |= term | trunc
        torch.cuda.synchronize()
        assert obs["observations"].shape == (n, cfg.observation_space_dim) and torch.isfinite(obs["observations"]).all()
        assert torch.isfinite(rew).all() and torch.isfinite(g["robot_state_tensor"]).all()
        img = g["depth_range_pixels"]
        assert img.shape == (n, 1, 135, 240) and float(img.max()) <= 1.0 and float(img.min()) >= -1.0 and (img > 0).any()
        # per-env gains were drawn inside the reset, between the configured bounds (randomize_params = True)
        kr = ctrl.K_rot_tensor_current.cpu().numpy()
        assert kr[:, 0].min() >= 1.6 - 1e-6 and kr[:, 0].max() <= 1.85 + 1e-6 and kr[:, 0].std() > 0
        # the velocity controller holds altitude within centimetres over 8 steps of 10 sub-steps (a 1.24 kg airframe on 4 x 10 N motors)
        alive = ~ever_reset  # (an env that crashed was re-placed; disturbances of up to 4.75 N are on for this robot)
        assert int(alive.sum()) >= n // 2 and float((g["robot_position"][:, 2] - z0)[alive].abs().max()) < 0.5
    finally:
        cfg.robot_name, cfg.controller_name, cfg.device, cfg.args = old


def test_root_link_quad_position_task_steps():
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    old = (cfg.robot_name, cfg.controller_name, cfg.device, cfg.args)
    cfg.robot_name, cfg.controller_name, cfg.device, cfg.args = "base_quad_root_link_control", "lee_attitude_control", DEV, {}
    try:
        n = 128
        task = task_registry.make_task("position_setpoint_task", seed=3, num_envs=n, headless=True)
        assert task.sim_env.robot_manager.robot.params_dict["root_link_mode"] == 1
        task.reset()
        for _ in range(20):
            obs, rew, term, trunc, info = task.step(torch.zeros(n, 4, device=DEV))
        torch.cuda.synchronize()
        assert torch.isfinite(obs["observations"]).all() and torch.isfinite(rew).all()
    finally:
        cfg.robot_name, cfg.controller_name, cfg.device, cfg.args = old


def test_compute_observations_and_randomize_params_stand_alone(orc):
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import navigation_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    old = (cfg.robot_name, cfg.controller_name, cfg.device, cfg.args)
    cfg.robot_name, cfg.controller_name, cfg.device, cfg.args = "lmf2_with_camera_64x48", "lmf2_velocity_control", DEV, {}
    try:
        n = 64
        task = task_registry.make_task("navigation_task", seed=4, num_envs=n, headless=True)
        env = task.sim_env
        task.reset()
        g, sc = env.global_tensor_dict, env.scene
        robot = env.robot_manager.robot
        # ---- compute_observations(): put half of the robots inside an obstacle of their env, by hand
        nb = int(env._buffers.num_boxes)
        boxes = np.ascontiguousarray(sc.boxes_soa.cpu().numpy().reshape(nb, 11, n)[:, :10, :].transpose(2, 0, 1))  # [N, B, centre 3 | quat 4 | half 3]
        pos = g["robot_position"].cpu().numpy().copy()
        pos[: n // 2] = boxes[: n // 2, 0, 0:3]  # the centre of obstacle 0 of the env, wherever it currently is
        g["robot_position"][:] = torch.from_numpy(pos).to(DEV)
        env.reset_tensors()
        pre = torch.zeros(n, dtype=torch.bool, device=DEV)
        pre[-3:] = True  # flags that are already set stay set (`+=` in the reference)
        g["crashes"][:] = pre
        env.compute_observations()
        torch.cuda.synchronize()
        got = g["crashes"].cpu().numpy().astype(bool)
        state = np.ascontiguousarray(g["robot_state_tensor"].cpu().numpy())
        want = pre.cpu().numpy().astype(np.uint8)
        orc.collide_sphere_boxes(robot.params_dict["collision_radius"], state, boxes, want)
        assert np.array_equal(got, want.astype(bool))
        assert got[-3:].all() and got[: n // 2].all()
        # ---- randomize_params(env_ids) outside a reset
        ctrl = robot.controller
        before = [t.clone() for t in (ctrl.K_pos_tensor_current, ctrl.K_linvel_tensor_current, ctrl.K_rot_tensor_current, ctrl.K_angvel_tensor_current)]
        ids = torch.tensor([1, 5, 17, 40], device=DEV)
        ctrl.randomize_params(ids)
        torch.cuda.synchronize()
        after = (ctrl.K_pos_tensor_current, ctrl.K_linvel_tensor_current, ctrl.K_rot_tensor_current, ctrl.K_angvel_tensor_current)
        others = torch.ones(n, dtype=torch.bool, device=DEV)
        others[ids] = False
        lo, hi = np.array(ctrl.gains_min, np.float32), np.array(ctrl.gains_max, np.float32)
        changed = 0
        for kk, (b, a) in enumerate(zip(before, after)):
            assert torch.equal(b[others], a[others])
            av = a[ids].cpu().numpy()
            l, h = np.minimum(lo[3 * kk:3 * kk + 3], hi[3 * kk:3 * kk + 3]), np.maximum(lo[3 * kk:3 * kk + 3], hi[3 * kk:3 * kk + 3])
            assert (av >= l - 1e-6).all() and (av <= h + 1e-6).all()
            changed += int((a[ids] != b[ids]).any())
        assert changed >= 2  # (K_pos has min == max: redrawn to the same value)
        # the step after it runs on the new gains without complaint
        task.step(torch.zeros(n, 4, device=DEV))
        torch.cuda.synchronize()
    finally:
        cfg.robot_name, cfg.controller_name, cfg.device, cfg.args = old
