"""GPU: the strict-RNG position step without a stream synchronisation (csrc/agx_strict.hip).

The mode's contract is the reference's consumption of torch's generator: rand_like draws for all N envs per reset quantity, only on
steps on which some env resets (env_manager.py:364-375, IGE_env_manager.py:513-519, base_multirotor.py:177-205,
motor_model.py:140-154).  Two things are pinned here, bit for bit, against torch itself:
  * agx_torch_uniform_fill == the `uniform_` calls it stands for (numbers and generator offset), for small tensors (one element per
    thread), tensors beyond torch's grid cap (the float4 unrolling), ragged sizes, non-zero starting offsets, several seeds;
  * a task stepped through agx_position_task_step_strict == the same task stepped through the general strict path (the dispatcher
    calls, `.item()` on the flag): observations, rewards, flags, states, and the generator's state after every step."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fill(shapes, seed, warm):
    from aerial_gym_simulator_amd import _lib

    lib = _lib.load()
    torch.cuda.init()  # (default_generators is filled by the lazy initialisation)
    gen = torch.cuda.default_generators[0]
    props = torch.cuda.get_device_properties(0)
    torch.manual_seed(seed)
    for _ in range(warm):  # a generator that has been used: non-zero offset
        torch.rand(1000, device=DEV)
    off = gen.get_offset()
    mine = [torch.full(s, -1.0, device=DEV) for s in shapes]
    outs = (C.c_void_p * len(mine))(*[m.data_ptr() for m in mine])
    numel = (C.c_int64 * len(mine))(*[m.numel() for m in mine])
    after = C.c_uint64(0)
    _lib.check(lib.agx_torch_uniform_fill(len(mine), outs, numel, gen.initial_seed() & 0xFFFFFFFFFFFFFFFF, off, props.multi_processor_count,
                                          props.max_threads_per_multi_processor, C.byref(after), _lib.current_stream(DEV)), "agx_torch_uniform_fill")
    real = [torch.empty(s, device=DEV).uniform_(0.0, 1.0) for s in shapes]
    torch.cuda.synchronize()
    return mine, real, after.value, gen.get_offset()


@pytest.mark.parametrize("shapes,seed,warm", [
    ([(8192, 3), (8192, 3), (8192, 13), (8192, 4), (8192, 4), (8192, 4), (8192, 4)], 1, 0),     # the strict reset of configs[1]
    ([(64, 3), (64, 3), (64, 13), (64, 4), (64, 4), (64, 4)], 77, 3),                             # configs[0], no kT draw
    ([(1000, 13), (17,), (1, 1), (255,), (257,)], 123456789012345, 5),                              # ragged
    ([(1 << 21, 13), (1 << 21, 4), (600001,)], 9, 1),                                               # beyond the grid cap: float4 unrolling
    ([(3000000,), (524288,), (524289,)], 2 ** 40 + 3, 2),
])
def test_uniform_fill_reproduces_torch_uniform(shapes, seed, warm):
    mine, real, after, after_real = _fill(shapes, seed, warm)
    assert after == after_real, (after, after_real)
    for j, (a, b) in enumerate(zip(mine, real)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (j, shapes[j], int((a != b).sum()))
        assert float(a.min()) >= 0.0 and float(a.max()) < 1.0


def _run(n, steps, episode_len, general, seed=11):
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    old = (cfg.device, cfg.controller_name, cfg.episode_len_steps, cfg.args)
    cfg.device, cfg.controller_name, cfg.episode_len_steps, cfg.args = DEV, "lee_position_control", episode_len, {"strict_rng": True}
    try:
        task = task_registry.make_task("position_setpoint_task", seed=seed, num_envs=n, headless=True)
        assert task._strict is not None, "the strict fast path was not taken (self-check failed?)"
        if general:
            task._plan = None  # the general path: env.step + reward + post_reward_calculation_step, draws through torch's dispatcher
        task.reset()
        # episodes spread over a few phases: some steps have resets, most have none
        steps_t = task.sim_env.global_tensor_dict["sim_steps"]
        steps_t[:] = (torch.arange(n, device=DEV, dtype=torch.int32) % 4) * (episode_len // 4)
        gen = torch.cuda.default_generators[0]
        g = torch.Generator(device=DEV).manual_seed(5)
        rec = []
        for t in range(steps):
            a = torch.rand(n, 4, device=DEV, generator=g) * 2 - 1
            obs, rew, term, trunc, _ = task.step(a)
            rec.append((obs["observations"].clone(), rew.clone(), term.clone(), trunc.clone(),
                        task.sim_env.global_tensor_dict["robot_state_tensor"].clone(), gen.get_offset()))
        resets = int(task.sim_env.global_tensor_dict["episode_count"].sum()) - n
        return rec, resets
    finally:
        cfg.device, cfg.controller_name, cfg.episode_len_steps, cfg.args = old


@pytest.mark.parametrize("n", [8192, 100])
def test_strict_fast_step_equals_the_general_strict_path(n):
    steps, L = 130, 40
    fast, resets_fast = _run(n, steps, L, general=False)
    slow, resets_slow = _run(n, steps, L, general=True)
    assert resets_fast == resets_slow and resets_fast >= 2 * n
    offsets = [r[5] for r in fast]
    assert len(set(offsets)) > 5 and sum(1 for a, b in zip(offsets, offsets[1:]) if a == b) > steps // 2  # draws on some steps only
    for t, (f, s) in enumerate(zip(fast, slow)):
        for k, name in enumerate(("observations", "rewards", "terminations", "truncations", "robot_state_tensor")):
            a, b = f[k], s[k]
            same = torch.equal(a.view(torch.int32), b.view(torch.int32)) if a.is_floating_point() else torch.equal(a, b)
            assert same, (t, name)
        assert f[5] == s[5], (t, "generator offset", f[5], s[5])


def test_a_torch_whose_uniform_kernel_differs_falls_back_to_the_dispatcher(monkeypatch):
    """EnvManager.strict_draw_plan checks the fused draw launch against the real `uniform_` calls once, on a saved-and-restored
    generator state; when they disagree (here: a test double of agx_torch_uniform_fill that fills zeros -- what an upgrade of torch
    that changes its kernel would look like) the task warns, takes the general strict path (dispatcher calls, `.item()`), and steps
    exactly like it: same numbers as a run that never had the fast path."""
    from aerial_gym_simulator_amd import _lib

    lib = _lib.load()
    real = lib.agx_torch_uniform_fill

    def zeros(count, outs, numel, seed, offset, sm, mt, after, stream):
        after._obj.value = offset + 4 * count  # (the right offset, the wrong numbers: the buffers stay as they are)
        return 0

    monkeypatch.setattr(lib, "agx_torch_uniform_fill", zeros)
    with pytest.warns(UserWarning, match="does not reproduce this torch build"):
        slow, resets = _run_with_plan_check(expect_fast=False)
    monkeypatch.setattr(lib, "agx_torch_uniform_fill", real)
    ref, resets_ref = _run(100, 60, 20, general=True)
    assert resets == resets_ref and len(slow) == len(ref)
    for t, (a, b) in enumerate(zip(slow, ref)):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[5] == b[5], t


def _run_with_plan_check(expect_fast):
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    n, steps, episode_len = 100, 60, 20
    old = (cfg.device, cfg.controller_name, cfg.episode_len_steps, cfg.args)
    cfg.device, cfg.controller_name, cfg.episode_len_steps, cfg.args = DEV, "lee_position_control", episode_len, {"strict_rng": True}
    try:
        task = task_registry.make_task("position_setpoint_task", seed=11, num_envs=n, headless=True)
        assert (task._strict is not None) == expect_fast and task._plan is None
        task.reset()
        steps_t = task.sim_env.global_tensor_dict["sim_steps"]
        steps_t[:] = (torch.arange(n, device=DEV, dtype=torch.int32) % 4) * (episode_len // 4)
        gen = torch.cuda.default_generators[0]
        g = torch.Generator(device=DEV).manual_seed(5)
        rec = []
        for _ in range(steps):
            a = torch.rand(n, 4, device=DEV, generator=g) * 2 - 1
            obs, rew, term, trunc, _ = task.step(a)
            rec.append((obs["observations"].clone(), rew.clone(), term.clone(), trunc.clone(), None, gen.get_offset()))
        return rec, int(task.sim_env.global_tensor_dict["episode_count"].sum()) - n
    finally:
        cfg.device, cfg.controller_name, cfg.episode_len_steps, cfg.args = old
