"""CPU: C-ABI surface, host logic, API contract (no compute calls, no GPU)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
from conftest import ROOT, load_golden


def test_library_builds_loads_and_exports_every_declared_symbol():
    from aerial_gym_simulator_amd import _build, _lib

    path = _build.build_library()
    assert os.path.exists(path)
    header = open(os.path.join(ROOT, "include", "aerial_gym_hip.h")).read()
    declared = set(re.findall(r"\b(agx_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 18
    lib = ctypes.CDLL(path)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/aerial_gym_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    L = _lib.load()
    assert L.agx_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define AGX_ABI_VERSION (\d+)", header).group(1))
    # links only against the HIP runtime / libc: no torch, no python in the C ABI library
    needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
    libs = re.findall(r"Shared library: \[(.*?)\]", needed)
    assert any("amdhip64" in x for x in libs)
    assert not any(("torch" in x) or ("python" in x) or ("c10" in x) or ("rccl" in x) for x in libs), libs  # RCCL is bound at run time


def test_graft_entry_build_runs_on_cpu():
    """The driver's "does it build" check: hipcc cross-compile + oracle + import, no GPU needed."""
    import __graft_entry__ as entry

    assert os.path.exists(entry.build())


def test_struct_layouts_match_header():
    """sizeof of the ctypes mirrors == sizeof in C (compiled with gcc from the header)."""
    from aerial_gym_simulator_amd import _lib

    src = '#include <stdio.h>\n#include "aerial_gym_hip.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(AgxRobotParams), sizeof(AgxEnvBuffers), sizeof(AgxResetArgs));printf("%zu %zu %zu %zu\\n", sizeof(AgxTaskArgs), sizeof(AgxRangeLimits), sizeof(AgxPositionStepPlan), sizeof(AgxImuArgs));printf("%zu %zu %zu\\n", sizeof(AgxNavRobotSideArgs), sizeof(AgxRobotStepArgs), sizeof(AgxLinkFrames));printf("%zu\\n", sizeof(AgxStrictStepPlan));return 0;}'
    exe = "/tmp/agx_sizeof"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src, text=True, check=True)
    sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True).stdout.split()]
    assert sizes == [ctypes.sizeof(_lib.AgxRobotParams), ctypes.sizeof(_lib.AgxEnvBuffers), ctypes.sizeof(_lib.AgxResetArgs),
                     ctypes.sizeof(_lib.AgxTaskArgs), ctypes.sizeof(_lib.AgxRangeLimits), ctypes.sizeof(_lib.AgxPositionStepPlan),
                     ctypes.sizeof(_lib.AgxImuArgs), ctypes.sizeof(_lib.AgxNavRobotSideArgs), ctypes.sizeof(_lib.AgxRobotStepArgs),
                     ctypes.sizeof(_lib.AgxLinkFrames), ctypes.sizeof(_lib.AgxStrictStepPlan)]


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "aerial_gym_simulator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)
    assert "oracle" not in open(os.path.join(ROOT, "include", "aerial_gym_hip.h")).read().lower()


def test_registries_and_names():
    import aerial_gym_simulator_amd as ag

    for name in ("lee_position_control", "lee_velocity_control", "lee_attitude_control", "lee_rates_control",
                 "lee_acceleration_control", "no_control", "octarotor_velocity_control", "rov_fully_actuated_control"):
        assert name in ag.controller_registry.get_controller_names()
    for name in ("base_quadrotor", "base_octarotor", "base_quadrotor_with_camera"):
        assert name in ag.robot_registry.get_robot_names()
    assert set(ag.task_registry.get_task_names()) >= {"position_setpoint_task", "navigation_task"}
    assert ag.sim_config_registry.make_sim("base_sim").sim.dt == 0.01
    with pytest.raises(ValueError):
        ag.robot_registry.get_robot_class("nope")


def _make_position_task(n=16):
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.task.position_setpoint_task import PositionSetpointTask

    cfg.controller_name = "lee_position_control"
    return PositionSetpointTask(cfg, num_envs=n, device="cpu", seed=3)


def test_tensor_dict_contract_and_aliasing():
    t = _make_position_task(16)
    g = t.sim_env.get_obs()
    for key, shape in (("robot_position", (16, 3)), ("robot_orientation", (16, 4)), ("robot_linvel", (16, 3)),
                       ("robot_angvel", (16, 3)), ("robot_body_linvel", (16, 3)), ("robot_body_angvel", (16, 3)),
                       ("robot_vehicle_orientation", (16, 4)), ("robot_vehicle_linvel", (16, 3)),
                       ("robot_euler_angles", (16, 3)), ("robot_actions", (16, 4)), ("robot_prev_actions", (16, 4)),
                       ("crashes", (16,)), ("truncations", (16,)), ("env_bounds_min", (16, 3)), ("env_bounds_max", (16, 3)),
                       ("gravity", (16, 3)), ("robot_mass", (16,)), ("robot_inertia", (16, 3, 3)), ("robot_state_tensor", (16, 13))):
        assert tuple(g[key].shape) == shape, key
    assert g["crashes"].dtype == torch.bool and t.sim_env.sim_steps.dtype == torch.int32
    # reference-shaped views alias the SoA storage (callers mutate in place)
    g["robot_position"][5, 2] = 7.5
    assert g["robot_state_soa"][2, 5] == 7.5 and g["robot_state_tensor"][5, 2] == 7.5
    g["robot_state_tensor"][3, 10:13] = torch.tensor([1.0, 2.0, 3.0])
    assert torch.equal(g["robot_angvel"][3], torch.tensor([1.0, 2.0, 3.0]))
    assert t.terminations is g["crashes"] and t.truncations is g["truncations"]
    assert abs(float(g["robot_mass"][0]) - 0.25) < 1e-7
    assert t.task_obs["observations"].shape == (16, 13) and t.task_obs["observations"].is_contiguous()
    assert t.action_space.shape == (4,) and t.observation_space["observations"].shape == (13,)


def test_robot_model_matches_urdf_derived_constants():
    from aerial_gym_simulator_amd.config.robot_config import BaseOctarotorCfg, BaseQuadCfg
    from aerial_gym_simulator_amd.robots.robot_model import composite_body, motor_wrench_map

    for name, cfg in (("quad", BaseQuadCfg), ("octarotor", BaseOctarotorCfg)):
        g = load_golden("robot_" + name)  # computed from the reference's URDF files
        m, com, J = composite_body(cfg.robot_model)
        assert abs(m - float(g["mass"])) < 1e-12 and np.abs(com - g["com"]).max() < 1e-12
        assert np.abs(J - g["inertia"]).max() < 1e-12
        ca = cfg.control_allocator_config
        W = motor_wrench_map(cfg.robot_model, ca.motor_directions, ca.motor_model_config.thrust_to_torque_ratio, com)
        assert np.abs(W - g["wrench_map"]).max() < 1e-12
        assert abs(cfg.robot_model.collision_sphere_radius - float(g["collision_radius"])) < 1e-15


def test_stepping_without_hip_device_fails_loudly():
    t = _make_position_task(8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        t.step(torch.zeros(8, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        t.reset()
    from aerial_gym_simulator_amd import _lib

    with pytest.raises(RuntimeError, match="HIP device"):
        _lib.dptr(torch.zeros(4))


def test_exchange_entry_points_reject_bad_arguments_without_a_gpu():
    """agx_exchange_* argument checks and the run-time RCCL binding fail with a message, not a crash; the
    process-group backend is the one a CPU job gets."""
    import ctypes as C

    from aerial_gym_simulator_amd import _lib
    from aerial_gym_simulator_amd.sharding import StepGather

    lib = _lib.load()
    uid = (C.c_char * 128)()
    assert lib.agx_exchange_unique_id(b"/nonexistent/librccl.so", uid, 128) != 0
    assert "cannot load RCCL" in lib.agx_last_error().decode()
    assert lib.agx_exchange_unique_id(None, uid, 64) != 0
    h = C.c_void_p()
    assert lib.agx_exchange_create(None, uid.raw, 128, 2, 2, 0, C.byref(h)) != 0 and h.value is None
    assert lib.agx_exchange_post(None, 0, None, None, 0, None, 0, None) != 0
    assert lib.agx_exchange_probe(None, None) < 0
    assert lib.agx_exchange_destroy(None) == 0
    sg = StepGather(4, 13, "cpu")
    assert sg.backend == "none" and sg.signal is None
    with pytest.raises(RuntimeError, match="HIP device"):
        StepGather(4, 13, "cpu", backend="rccl_thread")


def test_env_count_above_the_32_bit_offset_range_is_refused_without_a_gpu():
    """The one-lane kernels address a tensor's [<= 16][N] floats with 32-bit byte offsets (buffer accesses): an env count above
    2^26 per GPU is refused by the argument check of every entry point, before anything is launched."""
    from aerial_gym_simulator_amd import _lib

    lib = _lib.load()
    B = _lib.AgxEnvBuffers()
    assert lib.agx_update_states(B, (1 << 26) + 1, None) != 0
    assert "2^26" in lib.agx_last_error().decode()
    assert lib.agx_update_states(B, 0, None) != 0 and "num_envs must be > 0" in lib.agx_last_error().decode()


def test_scene_manager_semantics():
    from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

    env = SimBuilder().build_env("base_sim", "env_with_random_boxes", "base_quadrotor_with_camera_64x48", "lee_velocity_control",
                                 "cpu", num_envs=6)
    sc = env.scene
    assert (sc.num_assets, sc.num_tris, sc.keep_in_env_num) == (106, 1272, 6)
    sem = sc.asset_semantic_id.numpy()
    assert set(sem[0, :6]) == {9, 10, 11, 12, 13, 14}            # walls keep their fixed ids
    assert sem[0, 6] == 106 and sem[1, 6] == 100 + 106 + 6        # global counter from 100 (env_manager.py:147)
    assert sc.tri_local.shape == (6, 1272, 9) and sc.tri_seg.shape == (6, 1272)
    he = sc.half_extents.numpy()
    assert he[:, 6:].min() >= 0.05 - 1e-6 and he[:, 6:].max() <= 0.6 + 1e-6
    g = env.get_obs()
    assert g["depth_range_pixels"].shape == (6, 1, 48, 64) and g["segmentation_pixels"].dtype == torch.int32
    env2 = SimBuilder().build_env("base_sim", "env_with_obstacles", "base_quadrotor_with_camera", "lee_velocity_control", "cpu", num_envs=3)
    assert (env2.scene.num_assets, env2.scene.keep_in_env_num) == (44, 9)  # 3 panels + 35 objects + 6 walls


def test_sharded_scenes_are_slices_of_the_global_scene_set():
    """Rank r of a sharded run owns global envs [r*N, (r+1)*N): same boxes, same semantic ids
    as the un-sharded run's slice."""
    from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

    name = ("base_sim", "env_with_random_boxes", "base_quadrotor_with_camera_64x48", "lee_velocity_control", "cpu")
    full = SimBuilder().build_env(*name, num_envs=6).scene
    for rank in (0, 1, 2):
        part = SimBuilder().build_env(*name, num_envs=2, args={"shard_rank": rank}).scene
        sl = slice(2 * rank, 2 * rank + 2)
        assert torch.equal(part.asset_semantic_id, full.asset_semantic_id[sl])
        assert torch.equal(part.half_extents, full.half_extents[sl])
        assert torch.equal(part.tri_seg, full.tri_seg[sl])


def test_navigation_action_transform_matches_reference():
    from aerial_gym_simulator_amd.config.task_config import navigation_task_config as cfg

    g = load_golden("reward_navigation")
    out = cfg.action_transformation_function(torch.from_numpy(g["action_transform_in"]))
    assert np.array_equal(out.numpy(), g["action_transform_out"])  # the reference's outputs, bit for bit (same torch build)
    # the launch-saving form against the reference's wording, beyond the clamp range too
    a = torch.rand(50000, 4, generator=torch.Generator().manual_seed(3)) * 3.0 - 1.5
    assert torch.equal(cfg.action_transformation_function(a), cfg.action_transformation_function_as_written(a))


def _gather_worker(rank, world, port, ret):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from aerial_gym_simulator_amd.sharding import StepGather, shard_range

    total, obs_dim = 10, 13
    lo, hi = shard_range(total, rank, world)
    full_obs = torch.arange(total * obs_dim, dtype=torch.float32).view(total, obs_dim)
    sg = StepGather(hi - lo, obs_dim, "cpu")
    ids = torch.arange(total)

    def fill(step):  # what the observation kernel writes into rows[parity] of this rank's envs
        par = step & 1
        sg.pack(par, full_obs[lo:hi] + step, ids[lo:hi].float() * (step + 1), (ids[lo:hi] + step) % 2 == 0, (ids[lo:hi] + step) % 3 == 0)
        return par

    def expect(buf, step):
        obs, rew, term, trunc = sg.unpack(buf)
        return (torch.equal(obs, full_obs + step) and torch.equal(rew, ids.float() * (step + 1))
                and torch.equal(term, (ids + step) % 2 == 0) and torch.equal(trunc, (ids + step) % 3 == 0))

    ok = expect(sg.exchange(fill(0)), 0)  # synchronous: this step's rows of every rank
    ok = ok and sg.exchange(fill(1), overlap=True) is None  # pipelined: first call has nothing to hand back
    for step in range(2, 7):
        ok = ok and expect(sg.exchange(fill(step), overlap=True), step - 1)
    sg.flush()
    ok = ok and expect(sg.gathered[6 & 1], 6)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_sharded_gather_world2_gloo():
    """N > 1 path on CPU: 2 processes, gloo, one all-gather of the packed step outputs ==
    concatenation of the shards."""
    import torch.multiprocessing as mp

    from aerial_gym_simulator_amd.sharding import semantic_id_offset, shard_range

    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert semantic_id_offset(2, 8192, 106) == 2 * 8192 * 106
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert dict(ret) == {0: True, 1: True}


def test_tensor_dict_hooks_cover_every_access_path():
    """values() / items() / copy() / dict(g) / **g of the global tensor dict run the read hooks too (the lean step
    recomputes derived tensors on read); a refusing hook fails a direct read, not a bulk one."""
    from aerial_gym_simulator_amd.tensors import LeanStepRefused, TensorDict

    g = TensorDict(a=1, b=2, c=3)
    seen = []
    g.on_read("a", lambda k: seen.append(k))

    def refuse(k):
        raise LeanStepRefused("refused " + k)

    g.on_read("c", refuse)
    assert g["a"] == 1 and g.get("a") == 1 and seen == ["a", "a"]
    del seen[:]
    assert sorted(g.values()) == [1, 2, 3] and seen == ["a"]
    assert dict(g.items()) == {"a": 1, "b": 2, "c": 3} and seen == ["a", "a"]
    assert g.copy() == {"a": 1, "b": 2, "c": 3} and len(seen) == 3
    import pytest

    with pytest.raises(RuntimeError):
        g["c"]
    with pytest.raises(RuntimeError):
        dict(g)  # dict(g) / **g go through __getitem__ (no silent bypass); the refusal is loud
    g2 = TensorDict(a=1)
    g2.on_read("a", lambda k: seen.append("g2"))
    assert dict(g2) == {"a": 1} and (lambda **kw: kw)(**g2) == {"a": 1} and seen[-2:] == ["g2", "g2"]
    # only the lean step's refusal is tolerated by the bulk paths; any other failure of a hook (a HIP error raised by
    # the refresh) propagates instead of handing out a stale tensor
    g3 = TensorDict(a=1)

    def broken(k):
        raise RuntimeError("hipErrorIllegalAddress")

    g3.on_read("a", broken)
    for bulk in (g3.values, g3.items, g3.copy):
        with pytest.raises(RuntimeError, match="hipErrorIllegalAddress"):
            bulk()


@pytest.mark.parametrize("robot,controller,kind", [("base_quadrotor", "lee_position_control", "position"),
                                                    ("base_octarotor", "octarotor_velocity_control", "velocity"),
                                                    ("base_quad_root_link_control", "lee_position_control", "position")])
def test_robot_plugin_path_restated_by_the_oracle_agrees_with_the_fused_substep(robot, controller, kind):
    """Robot plug-in: per-body force / torque tensors as BaseMultirotor.step leaves them (orc.robot_step), reduced with
    the link-frame table of the config (robots.robot_model.link_frames) = the net body wrench the fused sub-step integrates
    (orc.substep: wrench_map folded ahead of time), to fp32 rounding -- the table and the map are two routes from the same URDF
    numbers.  CPU only: the oracle against itself and the host's table; the GPU test pins the kernels to these functions."""
    import oracle as orc

    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.registry.controller_registry import controller_registry
    from aerial_gym_simulator_amd.registry.robot_registry import robot_registry
    from aerial_gym_simulator_amd.registry.sim_registry import sim_config_registry
    from aerial_gym_simulator_amd.robots.robot_model import link_frames, robot_params_dict

    cfg = robot_registry.get_robot_config(robot)
    ccfg = controller_registry.get_controller_config(controller)
    pd = robot_params_dict(cfg, ccfg, kind, sim_config_registry.make_sim("base_sim"))
    P = orc.make_params(pd)
    n, M = 64, pd["num_motors"]
    rng = np.random.default_rng(3)
    state = np.zeros((n, 13), np.float32)
    state[:, 0:3] = rng.uniform(-1, 1, (n, 3))
    q = rng.normal(size=(n, 4))
    state[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    state[:, 7:13] = rng.uniform(-1, 1, (n, 6))
    thrust = rng.uniform(0, 1, (n, M)).astype(np.float32)
    kT = np.full((n, M), 1.2e-5, np.float32)
    tau = np.full((n, M), 0.03, np.float32)
    gains = [np.tile(((np.array(getattr(ccfg, f"K_{k}_tensor_max"), np.float32) + np.array(getattr(ccfg, f"K_{k}_tensor_min"), np.float32)) / 2), (n, 1))
             for k in ("pos", "vel", "rot", "angvel")]
    action = rng.uniform(-1, 1, (n, pd["num_actions"])).astype(np.float32)
    mask = [0] if pd["root_link_mode"] else [int(b) for b in cfg.control_allocator_config.application_mask]
    NB = max(int(b) for b in cfg.control_allocator_config.application_mask) + 1
    th_a, th_b = thrust.copy(), thrust.copy()
    o, F, T = orc.robot_step(P, state, action, th_a, kT, tau, tau, *gains, NB, mask)
    ref = orc.substep(P, state.copy(), action, th_b, kT, tau, tau, *gains, integrate=False)
    assert np.array_equal(th_a, th_b) and np.array_equal(o.wrench_cmd, ref.wrench_cmd) and np.array_equal(o.euler, ref.euler)
    L, known = link_frames(cfg, NB)
    rot = np.array([[L.rot[b][c] for c in range(9)] for b in range(NB)], np.float32)
    pos = np.array([[L.pos[b][c] for c in range(3)] for b in range(NB)], np.float32)
    net = orc.net_body_wrench(rot, pos, F, T)
    scale = np.abs(ref.body_wrench).max(axis=0) + 1e-3
    assert (np.abs(net - ref.body_wrench) / scale).max() < 2e-6
    nz = sorted(set(np.nonzero(np.abs(F).sum(axis=(0, 2)) + np.abs(T).sum(axis=(0, 2)))[0].tolist()))
    assert set(nz) <= set([0] + mask) and all(known[b] for b in nz)  # only bodies whose pose the table knows carry a wrench
