"""GPU: the lean step (args={"lean_step": True} -> AGX_LAUNCH_LEAN, batches above 65 536 envs): the fused position-task step
without the stores of the tensors that exist only to be looked at through the dict.  Everything the step PRODUCES --
state, motor thrusts, rewards, observations, flags, resets -- is bit-identical to the ordinary step; the derived tensors
are recomputed from the current state when a dict key is read; the action history is refused."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_lean_step_is_bit_identical_and_derived_tensors_are_served_on_demand():
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    old = (cfg.episode_len_steps, cfg.args, cfg.device, cfg.controller_name)
    n = 70000  # above EnvManager.LEAN_MIN_ENVS: the one-lane kernels, where the flag is implemented
    try:
        tasks = []
        for lean in (False, True):
            cfg.device, cfg.episode_len_steps, cfg.controller_name = DEV, 7, "lee_position_control"
            cfg.args = {"rng_seed": 99, "lean_step": lean}
            t = task_registry.make_task("position_setpoint_task", seed=2, num_envs=n, headless=True)
            t.reset()
            tasks.append(t)
        eager, lean = tasks
        assert lean.sim_env._lean and not eager.sim_env._lean and lean.sim_env._buffers.launch_flags == 4
        g = torch.Generator(device=DEV).manual_seed(5)
        for step in range(20):
            a = torch.rand(n, 4, device=DEV, generator=g) * 2 - 1
            (o0, r0, te0, tr0, _), (o1, r1, te1, tr1, _) = [t.step(a) for t in tasks]
            assert torch.equal(o0["observations"], o1["observations"]), step
            assert torch.equal(r0, r1) and torch.equal(te0, te1) and torch.equal(tr0, tr1), step
            for key in ("robot_state_tensor",):
                assert torch.equal(dict.__getitem__(eager.obs_dict, key), dict.__getitem__(lean.obs_dict, key)), (step, key)
            mm0 = eager.sim_env.robot_manager.robot.control_allocator.motor_model.current_motor_thrust
            mm1 = lean.sim_env.robot_manager.robot.control_allocator.motor_model.current_motor_thrust
            assert torch.equal(mm0, mm1), step
        assert int(lean.sim_env.global_tensor_dict["episode_count"].sum()) >= 2 * n  # resets were part of it
        # a derived key read through the dict: recomputed from the CURRENT state (the eager task's copy is what the last kernel
        # left there; agx_update_states on it gives the same fresh values)
        e1 = lean.obs_dict["robot_euler_angles"].clone()
        w1 = lean.obs_dict["robot_body_angvel"].clone()
        eager.sim_env.update_states()
        assert torch.equal(e1, dict.__getitem__(eager.obs_dict, "robot_euler_angles"))
        assert torch.equal(w1, dict.__getitem__(eager.obs_dict, "robot_body_angvel"))
        with pytest.raises(RuntimeError, match="not maintained by the lean step"):
            lean.obs_dict["robot_prev_actions"]
        # and the step after a dict read is unaffected
        a = torch.rand(n, 4, device=DEV, generator=g) * 2 - 1
        (o0, r0, *_), (o1, r1, *_) = [t.step(a) for t in tasks]
        assert torch.equal(o0["observations"], o1["observations"]) and torch.equal(r0, r1)
    finally:
        cfg.episode_len_steps, cfg.args, cfg.device, cfg.controller_name = old


def test_lean_step_is_refused_where_it_cannot_hold():
    """below the size threshold the flag is simply not set (the four-lane kernels do not implement it); with the navigation
    reward (it reads the action history) the library refuses the launch"""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    old = (cfg.args, cfg.device)
    try:
        cfg.device, cfg.args = DEV, {"lean_step": True}
        t = task_registry.make_task("position_setpoint_task", seed=2, num_envs=4096, headless=True)
        assert not t.sim_env._lean and t.sim_env._buffers.launch_flags == 0
        t.reset()
        t.step(torch.zeros(4096, 4, device=DEV))
        assert t.obs_dict["robot_prev_actions"].shape == (4096, 4)
    finally:
        cfg.args, cfg.device = old


def test_lean_step_is_opt_in_at_every_batch_size():
    """round 5: the dict behaves the same at every num_envs unless the caller asks for the lean
    step.  A tensor reference cached BEFORE stepping stays finite and maintained above 65 536 envs; args={"lean_step": True} turns
    the lean step on there (and logs it), where a dict read recomputes into the tensor a cached reference points at."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    old = (cfg.args, cfg.device, cfg.controller_name)
    try:
        cfg.device, cfg.args, cfg.controller_name = DEV, {}, "lee_position_control"
        big = task_registry.make_task("position_setpoint_task", seed=2, num_envs=70000, headless=True)
        assert not big.sim_env._lean and big.sim_env._buffers.launch_flags == 0
        cached = big.obs_dict["robot_euler_angles"]
        big.reset()
        for _ in range(3):
            big.step(torch.rand(70000, 4, device=DEV) * 2 - 1)
        assert torch.isfinite(cached).all() and cached.abs().sum() > 0
        assert big.obs_dict["robot_prev_actions"].shape == (70000, 4)
        cfg.args = {"lean_step": True}
        lean = task_registry.make_task("position_setpoint_task", seed=2, num_envs=70000, headless=True)
        assert lean.sim_env._lean and lean.sim_env._buffers.launch_flags == 4
        raw = dict.__getitem__(lean.obs_dict, "robot_euler_angles")  # a reference taken behind the dict's back
        lean.reset()
        lean.step(torch.zeros(70000, 4, device=DEV))
        fresh = lean.obs_dict["robot_euler_angles"]  # a dict read recomputes from the current state, into the same tensor
        assert fresh.data_ptr() == raw.data_ptr() and torch.isfinite(fresh).all()
    finally:
        cfg.args, cfg.device, cfg.controller_name = old
