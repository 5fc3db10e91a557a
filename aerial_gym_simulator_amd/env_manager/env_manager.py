"""EnvManager: same orchestration API as aerial_gym/env_manager/env_manager.py, driving the
HIP library instead of Isaac Gym + Warp + torch op chains.

Per env step (`step` + the task's reward + `post_reward_calculation_step`) the launches are
    agx_dynamics_substeps   k sub-steps: controller .. integration .. collision   (1 kernel)
    agx_reward_*            task reward, crash / truncation flags, reset set       (1 kernel)
    agx_reset_masked        reset sampling + derived-state refresh                  (1 kernel)
    [scene + BVH rebuild of reset envs, sensor pose, ray-cast, post-processing]
    agx_obs_*               observation packing                                     (1 kernel)
all stream-ordered on torch's current stream, with no host synchronisation unless the
caller asks for the reset set as indices or `strict_rng` is requested.
"""
import math

import torch

from .. import _lib
from .._lib import AgxEnvBuffers, AgxResetArgs
from ..registry.env_registry import env_config_registry
from ..registry.robot_registry import robot_registry
from ..registry.sim_registry import sim_config_registry
from ..robots.robot_manager import RobotManagerHIP
from ..tensors import aos_view, soa, LeanStepRefused, TensorDict
from ..utils import roctx
from ..utils.logging import CustomLogger
from ..utils.random_source import TorchRandomSource
from .asset_manager import AssetManager
from .base_env_manager import BaseManager
from .scene_manager import SceneManager

logger = CustomLogger("env_manager")


class ResetSet:
    """Lazy stand-in for the `envs_to_reset` index tensor returned by the reference
    (env_manager.py:364-375).  Holding the device mask keeps the step free of host syncs;
    `len()`, iteration or `.indices` materialise the indices (one sync) on demand."""

    def __init__(self, mask):
        self.mask = mask
        self._idx = None

    @property
    def indices(self):
        if self._idx is None:
            self._idx = self.mask.nonzero(as_tuple=False).squeeze(-1)
        return self._idx

    def __len__(self):
        return int(self.indices.numel())

    def __iter__(self):
        return iter(self.indices)

    def __getitem__(self, i):
        return self.indices[i]


class EnvManager(BaseManager):
    def __init__(self, sim_name, env_name, robot_name, controller_name, device, args=None, num_envs=None,
                 use_warp=None, headless=None):
        self.robot_name, self.controller_name = robot_name, controller_name
        self.sim_config = sim_config_registry.make_sim(sim_name)
        super().__init__(env_config_registry.make_env(env_name), device)
        if num_envs is not None:
            self.cfg.env.num_envs = num_envs
        if use_warp is not None:
            self.cfg.env.use_warp = use_warp
        if headless is not None:
            self.sim_config.viewer.headless = headless
        self.num_envs = self.cfg.env.num_envs
        self.use_warp = self.cfg.env.use_warp
        self.env_args = args or {}
        # strict_rng: consume the torch RNG exactly when and how the reference does (needs a
        # host sync per step to learn whether any env resets).  Default: draw every step.
        self.strict_rng = bool(self.env_args.get("strict_rng", False))
        self.random_source = self.env_args.get("random_source") or TorchRandomSource(device)
        # seed of the device-side counter RNG used by the sync-free mode (Philox4x32-10)
        self.rng_seed = int(self.env_args.get("rng_seed", torch.initial_seed())) & 0xFFFFFFFFFFFFFFFF
        self._parity = 0
        self.task_args = None   # AgxTaskArgs: reward / flags fused into the env-step launch
        self.post_obs = None    # (target_ptr, obs_ptr): observation fused into the reset launch
        self.global_tensor_dict = TensorDict()
        self.keep_in_env = None
        self.step_counter = 0
        self._stream_cache = None
        self._step_counter_dev = None
        self._lib = None
        self.populate_env(env_cfg=self.cfg, sim_cfg=self.sim_config)
        self.prepare_sim()
        self.sim_steps = self.global_tensor_dict["sim_steps"]

    # ------------------------------------------------------------------ construction
    def populate_env(self, env_cfg, sim_cfg):
        N, dev, g = self.num_envs, self.device, self.global_tensor_dict
        g["random_source"] = self.random_source
        g["env_manager"] = self
        g["strict_rng"] = self.strict_rng
        g["sim_config"] = sim_cfg
        g["crashes"] = torch.zeros(N, dtype=torch.bool, device=dev)
        g["truncations"] = torch.zeros(N, dtype=torch.bool, device=dev)
        g["reset_mask"] = torch.zeros(N, dtype=torch.uint8, device=dev)
        g["reset_flag"] = torch.zeros(2, dtype=torch.int32, device=dev)  # double buffered by step parity
        g["episode_count"] = torch.zeros(N, dtype=torch.int32, device=dev)
        g["sim_steps"] = torch.zeros(N, dtype=torch.int32, device=dev)
        self.collision_tensor, self.truncation_tensor = g["crashes"], g["truncations"]
        self.num_env_actions = env_cfg.env.num_env_actions
        g["num_env_actions"] = self.num_env_actions
        g["env_actions"] = None
        g["prev_env_actions"] = None
        # root state (IGE_env_manager.py:293-358): SoA storage, reference-shaped aliases
        state = soa(13, N, dev)
        state[6] = 1.0
        g["robot_state_soa"] = state
        g["robot_state_tensor"] = aos_view(state)
        g["robot_position"] = aos_view(state, 0, 3)
        g["robot_orientation"] = aos_view(state, 3, 7)
        g["robot_linvel"] = aos_view(state, 7, 10)
        g["robot_angvel"] = aos_view(state, 10, 13)
        # env bounds (IGE_env_manager.py:150-166, 513-519)
        e = env_cfg.env
        self.bounds_soa = (soa(3, N, dev), soa(3, N, dev))
        lo = (torch.tensor(e.lower_bound_min) + torch.tensor(e.lower_bound_max)) / 2.0
        hi = (torch.tensor(e.upper_bound_min) + torch.tensor(e.upper_bound_max)) / 2.0
        self.bounds_soa[0][:] = lo.to(dev).unsqueeze(1)
        self.bounds_soa[1][:] = hi.to(dev).unsqueeze(1)
        g["env_bounds_min"], g["env_bounds_max"] = aos_view(self.bounds_soa[0]), aos_view(self.bounds_soa[1])
        g["gravity"] = torch.tensor(sim_cfg.sim.gravity, device=dev).expand(N, -1)
        g["dt"] = sim_cfg.sim.dt
        # obstacles: which boxes live in which env (asset_loader.py:148-194 semantics)
        # sharded runs: global index of this process's first env (sharding.shard_range()[0]).  `shard_rank` alone is enough
        # only when every rank holds the same number of envs; with remainders (10 envs on 3 ranks: 4, 3, 3) rank * N_local
        # would make neighbouring ranks share scene seeds, segmentation ids and device-RNG streams.
        rank = int(self.env_args.get("shard_rank", 0))
        self.env_offset = int(self.env_args.get("env_offset", rank * N))
        self.scene = SceneManager(env_cfg, N, dev, self.random_source, shard_rank=rank, env_offset=self.env_offset,
                                  box_objects=bool(self.env_args.get("bvh_box_objects", True)))
        self.keep_in_env = self.scene.keep_in_env_num
        g["num_obstacles_in_env"] = self.scene.num_assets
        self.robot_manager = RobotManagerHIP(self.global_tensor_dict, self.cfg, self.robot_name, self.controller_name, dev)

    def prepare_sim(self):
        g = self.global_tensor_dict
        self._lib = _lib.load()  # fails loudly when the HIP library is absent
        self.scene.prepare_for_simulation(g)
        self.asset_manager = AssetManager(g, self.keep_in_env, self.scene)
        self.robot_manager.prepare_for_sim(g, self.scene)
        self.num_robot_actions = g["num_robot_actions"]
        self._bind()
        self.asset_manager.prepare_for_sim(self)

    def _bind(self):
        """Collect raw device pointers once: tensors are allocated once and never re-allocated."""
        g, robot = self.global_tensor_dict, self.robot_manager.robot
        if torch.device(self.device).type != "cuda":
            # tensors can be inspected on the CPU (host-logic tests), stepping cannot
            self._buffers = None
            return
        mm = robot.control_allocator.motor_model
        B = AgxEnvBuffers()
        p = _lib.dptr
        B.state, B.derived = p(g["robot_state_soa"]), p(g["robot_derived_soa"])
        B.actions, B.prev_actions = p(g["robot_actions_soa"]), p(g["robot_prev_actions_soa"])
        B.motor_thrust, B.motor_kT = p(mm.thrust_soa), p(mm.kT_soa)
        # parameters that are not randomised (min == max in the config) travel as constants
        # in AgxRobotParams instead of [*, N] buffers: fewer HBM reads per env step
        ctrl, params = robot.controller, robot.params
        rng = mm.ranges
        uniform_tau = rng["tau_inc"][0] == rng["tau_inc"][1] and rng["tau_dec"][0] == rng["tau_dec"][1]
        uniform_tau = uniform_tau and not self.env_args.get("per_env_motor_constants", False)
        params.tau_inc_uniform, params.tau_dec_uniform = rng["tau_inc"][0], rng["tau_dec"][0]
        B.motor_tau_inc = None if uniform_tau else p(mm.tau_inc_soa)
        B.motor_tau_dec = None if uniform_tau else p(mm.tau_dec_soa)
        has_gains = g.get("controller_gains_soa") is not None
        per_env_gains = has_gains and (bool(getattr(ctrl.cfg, "randomize_params", False))
                                       or bool(self.env_args.get("per_env_gains", False)))
        if has_gains:
            import numpy as np

            for i in range(12):  # (max + min) / 2 in fp32, like K_*_tensor_current (base_lee_controller.py:59-62)
                params.gains_uniform[i] = float((np.float32(ctrl.gains_max[i]) + np.float32(ctrl.gains_min[i])) / np.float32(2.0))
            ctrl._per_env_gains_bound = per_env_gains
        B.gains = p(g["controller_gains_soa"]) if per_env_gains else None
        B.wrench_cmd = None  # only the stand-alone controller call stores the wrench
        B.crashes, B.truncations = p(g["crashes"]), p(g["truncations"])
        B.sim_steps, B.reset_mask, B.reset_flag = p(g["sim_steps"]), p(g["reset_mask"]), p(g["reset_flag"])
        B.flag_parity = self._parity
        B.episode_count = p(g["episode_count"])
        B.bounds_min, B.bounds_max = p(self.bounds_soa[0]), p(self.bounds_soa[1])
        B.disturb = None
        for i, v in enumerate(robot.max_force_and_torque_disturbance):
            B.disturb_max[i] = v
        B.disturb_prob = 0.0
        B.rng_seed = self.rng_seed
        B.env_index_base = self.env_offset
        B.step_counter = 0
        B.boxes = p(self.scene.boxes_soa) if self.scene.num_assets > 0 else None
        B.num_boxes = self.scene.num_prims  # collision boxes: one per primitive (= per asset for box scenes)
        imu = self.robot_manager.imu_sensor
        B.body_force = p(imu.body_force) if imu is not None else None
        self._buffers = B
        self._zero_wrench = None
        self._params = robot.params
        robot._env_binding = self
        robot.controller._env_binding = self
        self._make_reset_args()
        self._disturb_buf = None
        self._reward_fresh = self._obs_fresh = self._mask_fresh = False
        if (self.env_args.get("lean_step") and self.num_envs >= self.LEAN_MIN_ENVS and self._params.controller != 8  # 8: external controller
                and not getattr(robot, "external_robot", False)):
            self._enable_lean_step()

    def _make_reset_args(self):
        N, dev = self.num_envs, self.device
        robot, e = self.robot_manager.robot, self.cfg.env
        M = robot.cfg.control_allocator_config.num_motors
        ctrl = robot.controller
        self._randomize_gains = bool(getattr(ctrl.cfg, "randomize_params", False)) and hasattr(ctrl, "gains_soa")
        # one flat buffer -> one RNG launch per draw in the sync-free mode
        sizes = dict(bounds_lo=3, bounds_hi=3, state=13, gains=12, tau_inc=M, tau_dec=M, thrust=M, kT=M)
        self._u_flat = torch.zeros(N * sum(sizes.values()), device=dev)
        self._u = {}
        off = 0
        for name, c in sizes.items():
            self._u[name] = self._u_flat[off:off + N * c].view(N, c)
            off += N * c
        R = AgxResetArgs()
        p = _lib.dptr
        if self.strict_rng:  # host tensors; otherwise all NULL = device generator
            R.u_bounds_lo, R.u_bounds_hi, R.u_state = p(self._u["bounds_lo"]), p(self._u["bounds_hi"]), p(self._u["state"])
            R.u_gains = p(self._u["gains"]) if self._randomize_gains else None
            R.u_tau_inc, R.u_tau_dec = p(self._u["tau_inc"]), p(self._u["tau_dec"])
            R.u_thrust, R.u_kT = p(self._u["thrust"]), p(self._u["kT"])
        R.randomize_gains = int(self._randomize_gains)
        R.seed = self.rng_seed
        for i in range(3):
            R.lower_bound_min[i], R.lower_bound_max[i] = e.lower_bound_min[i], e.lower_bound_max[i]
            R.upper_bound_min[i], R.upper_bound_max[i] = e.upper_bound_min[i], e.upper_bound_max[i]
        for i in range(13):
            R.min_state[i], R.max_state[i] = robot.min_init_state[i], robot.max_init_state[i]
        if hasattr(ctrl, "gains_min"):
            for i in range(12):
                R.gains_min[i], R.gains_max[i] = ctrl.gains_min[i], ctrl.gains_max[i]
        rng = robot.control_allocator.motor_model.ranges
        R.tau_inc_min, R.tau_inc_max = rng["tau_inc"]
        R.tau_dec_min, R.tau_dec_max = rng["tau_dec"]
        R.kT_min, R.kT_max = rng["kT"]
        self._reset_args = R

    def enable_device_step_counter(self):
        """From now on the kernels read the env-step index from device memory (AgxEnvBuffers.step_counter_dev) instead of a
        kernel argument: a captured step can be replayed.  The caller advances it with agx_step_counter_advance as the
        last launch of every step (tasks do); the host copy `step_counter` keeps counting alongside."""
        self._require_device()
        if self._step_counter_dev is None:
            self._step_counter_dev = torch.tensor([self.step_counter & 0x7FFFFFFF], dtype=torch.int32, device=self.device)
            self._buffers.step_counter_dev = _lib.dptr(self._step_counter_dev)

    def bind_step_rows(self, rows, reward, signal=None):
        """Multi-GPU exchange rows (sharding.StepGather): `rows` [2, N, obs_dim + 3], one per step
        parity, written by the observation kernels next to the observation itself.  `signal` (int32 [4],
        zeros): the last wave of such a kernel stores signal[parity] = step_counter + 1 once every row
        is visible device-wide (AgxEnvBuffers.step_signal); the exchange waits on it on the device."""
        self._require_device()
        B = self._buffers
        B.step_rows[0], B.step_rows[1] = _lib.dptr(rows[0]), _lib.dptr(rows[1])
        B.step_reward = _lib.dptr(reward)
        B.step_signal = _lib.dptr(signal) if signal is not None else None
        self._step_rows = (rows, reward, signal)  # keep alive

    # ---- peer push of the exchange rows by the observation kernels themselves (sharding.StepGather, backend "peer_push") ----
    _push = None

    def bind_peer_push(self, recv_ptrs, flag_ptrs, rank, world, slots, row_len, reward, signal, timed_out_ptr):
        """recv_ptrs / flag_ptrs: the addresses IN THIS PROCESS of every rank's receive buffer [slots][world][N][row_len] and
        flag array [slots][world] (agx_exchange_push_peers).  From now on every step's rows are stored by the row-writing
        kernels into slot (seq - 1) % slots of EVERY rank's buffer (AgxEnvBuffers.push_*): no launch and no host call per
        step for the exchange; `_advance_push()` moves the sequence number once per env step."""
        self._require_device()
        if world > 8:
            raise ValueError("peer push from the kernels covers one node (at most 8 ranks)")
        B = self._buffers
        peers = [w for w in range(world) if w != rank]
        for j in range(7):
            B.push_delta[j] = (recv_ptrs[peers[j]] - recv_ptrs[rank]) if j < len(peers) else 0
        for w in range(8):
            B.push_flags[w] = flag_ptrs[w] if w < world else None
        B.push_world, B.push_rank = world, rank
        B.push_seq = B.push_wait_seq = 0
        B.push_flag_index = B.push_wait_index = 0
        B.push_timed_out = timed_out_ptr
        B.step_reward = _lib.dptr(reward)
        B.step_signal = _lib.dptr(signal)  # word [2]: the row-writing kernels' workgroup arrival counter
        B.push_base, B.push_slice_bytes, B.push_slots = recv_ptrs[rank], self.num_envs * row_len * 4, slots
        self._push = dict(rank=rank, world=world, slots=slots)
        B.step_rows[0] = B.step_rows[1] = recv_ptrs[rank] + rank * B.push_slice_bytes  # slot 0 until the first step
        self._step_rows = (None, reward, signal)  # keep alive

    def unbind_peer_push(self):
        if self._push is not None:
            B = self._buffers
            B.push_world = 0
            B.push_seq = B.push_wait_seq = 0
            B.step_rows[0] = B.step_rows[1] = None
            B.step_signal = None
            self._push = None

    def _advance_push(self):
        """new env step: the slot its rows go to, the sequence number that will announce them, and the step (two back) whose
        arrival from every rank the step waits for before anything is overwritten (agx_push_advance; the position task's
        one-call step does this inside the library)"""
        _lib.check(self._lib.agx_push_advance(self._buffers), "agx_push_advance")

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        """torch's current stream on this device.  Looked up once per public call (step / reset / render ...):
        torch.cuda.current_stream() costs microseconds and a navigation step makes ~15 launches."""
        s = self._stream_cache
        if s is None:
            s = self._stream_cache = _lib.current_stream(self.device)
        return s

    def _new_call(self):
        self._stream_cache = None
        self._derived_stale = True

    # ---- lean step (args={"lean_step": True}): AGX_LAUNCH_LEAN for batches far above 65 536 envs, where bytes matter ------
    LEAN_MIN_ENVS = 65537
    _lean = False
    _derived_stale = False

    def _enable_lean_step(self):
        """The fused step stops maintaining the tensors that exist only to be looked at through the dict (Euler angles, vehicle
        quaternion / velocity, robot_actions, robot_prev_actions: 88 of the ~330 bytes an env moves per step).  Reading a
        derived key recomputes all of them from the CURRENT state (agx_update_states) -- fresh values, where the eagerly
        maintained ones are one sub-step stale like the reference's; the action history is refused."""
        g = self.global_tensor_dict
        self._lean = True
        self._buffers.launch_flags = 4
        logger.warning("lean step enabled (args={'lean_step': True}, %d envs): robot_euler_angles / robot_vehicle_orientation / "
                       "robot_vehicle_linvel are recomputed from the CURRENT state when read through the dict (tensor references held "
                       "across steps show NaN until then); robot_actions / robot_prev_actions are not maintained." % self.num_envs)

        def refresh(_key):
            if self._derived_stale and self._lean:
                self._derived_stale = False
                self.update_states()

        def refuse(key):
            if self._lean:
                raise LeanStepRefused(f"'{key}' is not maintained by the lean step (args={{'lean_step': True}}); the task keeps the action "
                                   "tensors it was handed (task.actions / task.prev_actions)")

        for key in ("robot_euler_angles", "robot_vehicle_orientation", "robot_vehicle_linvel", "robot_body_linvel", "robot_body_angvel"):
            g.on_read(key, refresh)
        for key in ("robot_actions", "robot_prev_actions"):
            g.on_read(key, refuse)
        # A tensor reference taken from the dict BEFORE this point (or kept across steps) bypasses the hooks: the tensors the lean
        # step no longer maintains are poisoned here, so that such a reference shows NaN -- until the next reset or dict read
        # refreshes them -- instead of plausible numbers of an earlier time (re-fetch through the dict each step;
        # the body-frame velocities stay maintained).
        for key in ("robot_euler_angles", "robot_vehicle_orientation", "robot_vehicle_linvel", "robot_actions", "robot_prev_actions"):
            t = dict.get(g, key)
            if t is not None and t.is_floating_point():
                t.fill_(float("nan"))
        self._derived_stale = True

    def _require_device(self):
        if self._buffers is None:
            raise RuntimeError(
                f"EnvManager on device '{self.device}': simulation needs a HIP device (cuda:N); "
                "aerial_gym_simulator_amd has no CPU fallback"
            )

    def update_states(self):
        self._require_device()
        _lib.check(self._lib.agx_update_states(self._buffers, self.num_envs, self._stream()), "agx_update_states")

    def controller_wrench(self, action):
        self._require_device()
        a = action.to(dtype=torch.float32).contiguous()
        B = self._buffers
        B.wrench_cmd = _lib.dptr(self.global_tensor_dict["controller_wrench_soa"])
        try:
            _lib.check(self._lib.agx_controller_wrench(self._params, B, self.num_envs, _lib.dptr(a), self._stream()),
                       "agx_controller_wrench")
        finally:
            B.wrench_cmd = None
        return self.robot_manager.robot.controller.wrench_command

    # ------------------------------------------------------------------ reset
    def _draw_reset_randoms(self, env_ids=None):
        """Uniform draws for the robot reset.  strict_rng reproduces the reference's calls:
        rand_like for all N in the order bounds(lo, hi) [IGE_env_manager.py:513-519] ...
        robot state [base_multirotor.py:182], gains [base_lee_controller.py:101-118, len(env_ids)
        rows], motor tau_inc, tau_dec, thrust, kT [motor_model.py:140-154]."""
        rs, u = self.random_source, self._u
        if not self.strict_rng:
            return  # the reset kernels draw with the device generator
        robot = self.robot_manager.robot
        rs.rand_into(u["bounds_lo"], tag="bounds_lo")
        rs.rand_into(u["bounds_hi"], tag="bounds_hi")
        self.asset_manager.draw_reset_randoms(env_ids)
        rs.rand_into(u["state"], tag="robot_state")
        if self._randomize_gains:
            n = len(env_ids)
            for k in range(4):
                u["gains"][env_ids, 3 * k:3 * k + 3] = rs.rand(n, 3, tag="gains%d" % k)
        rs.rand_into(u["tau_inc"], tag="tau_inc")
        rs.rand_into(u["tau_dec"], tag="tau_dec")
        rs.rand_into(u["thrust"], tag="thrust")
        if robot.cfg.control_allocator_config.motor_model_config.use_rps:
            rs.rand_into(u["kT"], tag="kT")
        self.robot_manager.draw_sensor_reset_randoms(env_ids)

    def strict_draw_plan(self):
        """Strict mode, plain case (no obstacles, sensors, per-env gains; the default torch generator): the tensors the per-step
        reset draws, in the reference's call order (_draw_reset_randoms above), for ONE fused launch that reproduces those
        `uniform_` calls (agx_torch_uniform_fill) -- or None where the calls must go through torch (a caller's random source /
        generator, draws shaped by the reset set, or a torch whose kernels the restatement does not match: checked here, once,
        against the real calls on a saved-and-restored generator state)."""
        rs = self.random_source
        sensor = self.robot_manager.warp_sensor
        if (not self.strict_rng or type(rs) is not TorchRandomSource or rs.generator is not None or self._randomize_gains
                or self.scene.num_assets > 0 or sensor is not None or self.robot_manager.imu_sensor is not None
                or self.robot_manager.robot.cfg.disturbance.enable_disturbance):
            return None
        names = ["bounds_lo", "bounds_hi", "state", "tau_inc", "tau_dec", "thrust"]
        if self.robot_manager.robot.cfg.control_allocator_config.motor_model_config.use_rps:
            names.append("kT")
        bufs = [self._u[k] for k in names]
        index = torch.device(self.device).index
        index = index if index is not None else torch.cuda.current_device()
        gen = torch.cuda.default_generators[index]
        props = torch.cuda.get_device_properties(index)
        plan = {"tensors": bufs, "generator": gen, "sm_count": int(props.multi_processor_count),
                "max_threads_per_sm": int(props.max_threads_per_multi_processor)}
        # self-check: the fused launch against the dispatcher calls it stands for, on this torch, this device, these shapes
        state = gen.get_state()
        try:
            seed, off = gen.initial_seed(), gen.get_offset()
            real = []
            for b in bufs:
                real.append(torch.empty_like(b).uniform_(0.0, 1.0))
            off_real = gen.get_offset()
            mine = [torch.full_like(b, -1.0) for b in bufs]
            C = _lib.C
            outs = (C.c_void_p * len(bufs))(*[m.data_ptr() for m in mine])
            numel = (C.c_int64 * len(bufs))(*[m.numel() for m in mine])
            after = C.c_uint64(0)
            _lib.check(self._lib.agx_torch_uniform_fill(len(bufs), outs, numel, seed & 0xFFFFFFFFFFFFFFFF, off, plan["sm_count"],
                                                        plan["max_threads_per_sm"], C.byref(after), self._stream()), "agx_torch_uniform_fill")
            ok = after.value == off_real and all(torch.equal(a, b) for a, b in zip(real, mine))
        finally:
            gen.set_state(state)
        if not ok:
            import warnings

            warnings.warn("agx_torch_uniform_fill does not reproduce this torch build's uniform_ kernel: the strict step draws through "
                          "torch's dispatcher (slower, same numbers)")
            return None
        return plan

    # A navigation task may hand over what follows the robot reset on the robot side of its step -- sensor mounts and target of
    # the envs that reset, the world pose of every sensor -- to run in the SAME launch (agx_nav_robot_side: one launch instead
    # of four dependent ones).  Only the per-step reset of task.step() takes it; an explicit reset_idx() stays as it was.
    _nav_side = None
    _sensor_pose_fresh = False    # the fused launch computed this step's sensor poses: HipSensor.compose_pose skips once
    _targets_reset_fused = False  # ... and resampled the targets of the reset envs: the task's _reset_targets skips once

    def enable_fused_robot_side(self, nav_args):
        self._nav_side = nav_args

    def _launch_reset(self, with_obs=False, per_step=False):
        """Device side of EnvManager.reset_idx for the envs flagged in reset_mask (all kernels
        return immediately when reset_flag[parity] is 0)."""
        self.asset_manager.reset_masked(self)  # obstacle poses, scene triangles, BVH, boxes
        if per_step and self._nav_side is not None and self.post_obs is None and not self.strict_rng:
            _lib.check(self._lib.agx_nav_robot_side(self._params, self._buffers, self.num_envs, self._reset_args,
                                                    _lib.C.byref(self._nav_side), self._stream()), "agx_nav_robot_side")
            self._sensor_pose_fresh = self._nav_side.num_sensors > 0
            self._targets_reset_fused = bool(self._nav_side.reset_target)
            return
        if with_obs and self.post_obs is not None:
            _lib.check(self._lib.agx_post_step_position(self._params, self._buffers, self.num_envs, self._reset_args,
                                                        self.post_obs[0], self.post_obs[1], self._stream()),
                       "agx_post_step_position")
            self._obs_fresh = True
        else:
            _lib.check(self._lib.agx_reset_masked(self._params, self._buffers, self.num_envs, self._reset_args, self._stream()),
                       "agx_reset_masked")
        self.robot_manager.reset_sensors_masked()

    @roctx.ranged("EnvManager.reset_idx")
    def reset_idx(self, env_ids=None):
        self._require_device()
        self._new_call()
        g = self.global_tensor_dict
        if env_ids is None:
            env_ids = torch.arange(self.num_envs, device=self.device)
        env_ids = torch.as_tensor(env_ids, device=self.device, dtype=torch.long)
        if env_ids.numel() == 0:
            return
        g["reset_mask"].zero_()
        g["reset_mask"][env_ids] = 1
        g["reset_flag"][self._parity] = 1
        self._draw_reset_randoms(env_ids)
        self._launch_reset()

    def reset_robots(self, env_ids):
        self.reset_idx(env_ids)

    def randomize_controller_gains(self, env_ids=None):
        """BaseLeeController.randomize_params(env_ids) called on its own (base_lee_controller.py:101-118): K_pos, K_linvel, K_rot,
        K_angvel of `env_ids` re-drawn uniformly between their configured bounds, four rand draws of [len(env_ids), 3] through the
        random source, in the reference's order (the tags of the strict reset path: gains0..3).  Inside a reset the same draw is
        part of agx_reset_masked (u_gains / the device stream): this entry point is for callers that randomise between resets.
        No-op unless the controller config has randomize_params (the reference returns early too)."""
        ctrl = self.robot_manager.robot.controller
        if not self._randomize_gains:
            return
        ids = torch.arange(self.num_envs, device=self.device) if env_ids is None else torch.as_tensor(env_ids, device=self.device, dtype=torch.long)
        if ids.numel() == 0:
            return
        lo, hi = torch.tensor(ctrl.gains_min, device=self.device), torch.tensor(ctrl.gains_max, device=self.device)
        for k, cur in enumerate((ctrl.K_pos_tensor_current, ctrl.K_linvel_tensor_current, ctrl.K_rot_tensor_current, ctrl.K_angvel_tensor_current)):
            u = self.random_source.rand(int(ids.numel()), 3, tag="gains%d" % k)
            cur[ids] = (hi[3 * k:3 * k + 3] - lo[3 * k:3 * k + 3]) * u + lo[3 * k:3 * k + 3]  # torch_rand_float_tensor: (upper - lower) * rand + lower

    def reset(self):
        self.reset_idx(torch.arange(self.num_envs, device=self.device))

    def reset_terminated_and_truncated_envs(self):
        g = self.global_tensor_dict
        if not self._mask_fresh:
            # nobody produced this step's reset set on the device (no fused task epilogue, no agx_reward_* call): derive
            # it from the flags as they stand, like the reference does right here (env_manager.py:364-371) -- stand-alone
            # EnvManager loops (examples/benchmark.py) and user tasks that set `truncations[:]` in torch depend on it
            _lib.check(self._lib.agx_reset_set(self._buffers, self.num_envs, int(self.cfg.env.reset_on_collision), self._stream()),
                       "agx_reset_set")
            self._mask_fresh = True
        if self.strict_rng and int(g["reset_flag"][self._parity].item()) != 0:  # host sync, like the reference's nonzero()/len()
            # the indices themselves (a second synchronisation) only where a draw is shaped by them: per-env controller gains, the
            # obstacles' half-resample, sensor mounts -- the plain position task draws for all N envs whatever the set is
            sensor = self.robot_manager.warp_sensor
            needs_ids = (self._randomize_gains or (self.scene.num_assets > 0 and int(g["num_obstacles_in_env"]) > 0)
                         or (sensor is not None and sensor.cfg.randomize_placement))
            env_ids = g["reset_mask"].nonzero(as_tuple=False).squeeze(-1) if needs_ids else None
            self._draw_reset_randoms(env_ids)
        self._launch_reset(with_obs=True, per_step=True)
        return ResetSet(g["reset_mask"])

    # ------------------------------------------------------------------ stepping
    def reset_tensors(self):
        self.collision_tensor[:] = 0
        self.truncation_tensor[:] = 0

    def num_physics_steps(self):
        e = self.cfg.env
        return max(math.floor(self.random_source.gauss(e.num_physics_steps_per_env_step_mean,
                                                       e.num_physics_steps_per_env_step_std)), 0)

    def _draw_disturbance(self, k):
        """apply_disturbance draws per sub-step (base_multirotor.py:213-234): bernoulli(p) [N],
        then two rand_like [N,3]; stored SoA [k][7][N] for the kernel."""
        robot = self.robot_manager.robot
        if not robot.cfg.disturbance.enable_disturbance or k == 0:
            self._buffers.disturb = None
            return
        if not self.strict_rng:  # drawn inside the kernel (Philox keyed by env, step, sub-step)
            self._buffers.disturb = None
            self._buffers.disturb_prob = float(robot.cfg.disturbance.prob_apply_disturbance)
            self._buffers.step_counter = self.step_counter & 0x7FFFFFFF
            return
        N, rs = self.num_envs, self.random_source
        if self._disturb_buf is None or self._disturb_buf.shape[0] < k:
            self._disturb_buf = torch.zeros(max(k, 1), 7, N, device=self.device)
        p = robot.cfg.disturbance.prob_apply_disturbance
        for s in range(k):
            self._disturb_buf[s, 0] = rs.bernoulli(p, N, tag="disturb_occ")
            self._disturb_buf[s, 1:4] = rs.rand(N, 3, tag="disturb_f").t()
            self._disturb_buf[s, 4:7] = rs.rand(N, 3, tag="disturb_t").t()
        self._buffers.disturb = _lib.dptr(self._disturb_buf)

    def simulate(self, actions, env_actions=None, num_substeps=1):
        """k physics sub-steps (pre_physics_step + physics_step + post_physics_step +
        compute_observations of the reference) in one launch."""
        self._require_device()
        a = actions
        if a.dtype != torch.float32 or not a.is_contiguous():
            a = a.to(dtype=torch.float32).contiguous()
        if a.shape != (self.num_envs, self.num_robot_actions):
            raise ValueError("Action tensor does not have the correct number of environments")
        self._draw_disturbance(num_substeps)
        if getattr(self.robot_manager.robot, "external_robot", False):
            self._simulate_with_external_robot(a, num_substeps)
        elif getattr(self.robot_manager.robot, "external_controller", False):
            self._simulate_with_external_controller(a, num_substeps)
        else:
            _lib.check(
                self._lib.agx_env_step(self._params, self._buffers, self.num_envs, _lib.dptr(a), num_substeps, self.task_args,
                                       self._stream()),
                "agx_env_step",
            )
        self._reward_fresh = self._mask_fresh = self.task_args is not None
        self._obs_fresh = False
        self.robot_manager.post_physics_step(num_substeps)

    def _simulate_with_external_controller(self, a, k):
        """A controller class the user registered (torch code, controller_registry.register_controller): per physics
        sub-step, exactly the reference's order -- pre_physics_step copies the action (robot_manager.py:486-489),
        BaseMultirotor.step refreshes the derived state tensors, clips the action to +-10 and calls the controller
        (base_multirotor.py:296-307) -- then ONE launch takes the returned wrench through allocation, motor model, drag,
        disturbance, integration and the collision test.  Crash flags accumulate over the launches; the step counter,
        truncation and the task's fused reward run with the last one (AgxEnvBuffers.launch_flags)."""
        g, B, robot = self.global_tensor_dict, self._buffers, self.robot_manager.robot
        try:
            for sub in range(max(k, 1)):
                B.launch_flags = (1 if sub > 0 else 0) | (2 if sub < k - 1 else 0) | (sub << 8)
                if k > 0:
                    g["robot_prev_actions"][:] = g["robot_actions"]
                    g["robot_actions"][:] = a
                    self.update_states()
                    wrench = robot.controller(torch.clamp(a, -10.0, 10.0))
                    if wrench.shape != (self.num_envs, 6):
                        raise ValueError(f"controller returned {tuple(wrench.shape)}, expected ({self.num_envs}, 6): [fx fy fz tx ty tz]")
                    w = wrench.to(dtype=torch.float32).contiguous()
                else:  # no physics sub-step: the kernel still reads actions_in as [N][6] (the values are unused)
                    if self._zero_wrench is None:
                        self._zero_wrench = torch.zeros(self.num_envs, 6, dtype=torch.float32, device=a.device)
                    w = self._zero_wrench
                _lib.check(self._lib.agx_env_step(self._params, B, self.num_envs, _lib.dptr(w), min(k, 1), self.task_args, self._stream()),
                           "agx_env_step")
        finally:
            B.launch_flags = 0

    # ---- robot plug-in (robots/base_robot.py:10-63, robot_manager.py:486-489) -------------------------------------
    _robot_step_args = None
    _robot_substep = 0
    _in_external_simulate = False

    def _robot_plugin_state(self):
        """per-body tensors, application mask and link-frame table of the robot plug-in path (built on first use)"""
        if self._robot_step_args is None:
            from ..robots.robot_model import link_frames

            g, robot = self.global_tensor_dict, self.robot_manager.robot
            R = _lib.AgxRobotStepArgs()
            F, T = g["robot_force_tensor"], g["robot_torque_tensor"]
            R.force, R.torque, R.num_bodies = _lib.dptr(F), _lib.dptr(T), int(F.shape[1])
            mask = [0] if robot.params_dict["root_link_mode"] else [int(b) for b in robot.cfg.control_allocator_config.application_mask]
            for j, b in enumerate(mask):
                R.body_of_motor[j] = b
            self._robot_step_args = R
            self._link_frames, known = link_frames(robot.cfg, int(F.shape[1]))
            self._unknown_bodies = [b for b, k in enumerate(known) if not k]
            self._net_wrench = torch.zeros(self.num_envs, 6, dtype=torch.float32, device=self.device)
            # a wrench on a body whose pose the robot_model table does not know cannot be placed: every sub-step ORs "such a body
            # carries a wrench" into this device flag (no synchronisation), read on the first step and every 32nd after it
            self._unknown_wrench = torch.zeros((), dtype=torch.bool, device=self.device)
            self._unknown_steps = 0
        return self._robot_step_args

    def _body_wrench_params(self):
        """the integrating launch of the plug-in path takes the net wrench as it is: the robot's CURRENT constants (robot.params
        may have been edited since the last step), controller id 'wrench'"""
        import copy

        P = copy.copy(self._params)
        P.controller = _lib.CTRL_IDS["wrench"]
        return P

    def _check_unknown_bodies(self, force=False):
        self._unknown_steps += 1
        if self._unknown_bodies and (force or self._unknown_steps == 1 or self._unknown_steps % 32 == 0) and bool(self._unknown_wrench):
            raise NotImplementedError(
                f"robot.step() wrote a wrench on bodies {self._unknown_bodies}, whose poses the config's robot_model does not "
                "give: add them as robot_model.link_xyz / link_rpy = {body index: [x, y, z] / [r, p, y]}")

    def robot_step(self, action):
        """BaseMultirotor.step(action) of the reference (base_multirotor.py:296-307) as one launch, agx_robot_step: derived
        tensors, motor thrusts and the per-body force / torque tensors as the reference's robot leaves them."""
        self._require_device()
        robot = self.robot_manager.robot
        if getattr(robot, "external_controller", False):
            raise RuntimeError("BaseMultirotor.step with an external controller class: call the controller yourself and write "
                               "robot_force_tensors / robot_torque_tensors in your step()")
        a = action
        if a.dtype != torch.float32 or not a.is_contiguous():
            a = a.to(dtype=torch.float32).contiguous()
        if a.shape != (self.num_envs, self.num_robot_actions):
            raise ValueError("Action tensor does not have the correct number of environments")
        if not self._in_external_simulate:  # BaseMultirotor.step called by hand: a call of its own (stream lookup, derived tensors)
            self._new_call()
        R = self._robot_plugin_state()
        R.substep = int(self._robot_substep)
        _lib.check(self._lib.agx_robot_step(self._params, self._buffers, self.num_envs, _lib.dptr(a), _lib.C.byref(R), self._stream()),
                   "agx_robot_step")

    def _simulate_with_external_robot(self, a, k):
        """A robot class the user registered (robot_registry.register) that overrides step(): per physics sub-step, the
        reference's order -- RobotManagerIGE.pre_physics_step copies the action and calls robot.step(actions)
        (robot_manager.py:486-489), which leaves each body's wrench (in that body's frame) in robot_force_tensors /
        robot_torque_tensors; Isaac Gym applies them and steps PhysX (IGE_env_manager.py:444-449, 477).  Here: the user's
        step() on the host (BaseMultirotor.step, reachable through super(), is ONE launch), agx_net_body_wrench reduces the
        per-body tensors to the net wrench on the rigid composite, and ONE launch integrates it and tests for collisions
        (AGX_CTRL_WRENCH + AGX_LAUNCH_BODY_WRENCH: no allocation / motor model / drag / disturbance there -- step() did
        whatever it does about them).  Flags, step counter, truncation and the task's fused reward run with the last launch."""
        g, B, robot = self.global_tensor_dict, self._buffers, self.robot_manager.robot
        self._robot_plugin_state()
        F, T = g["robot_force_tensor"], g["robot_torque_tensor"]
        P_wrench = self._body_wrench_params()
        self._in_external_simulate = True
        try:
            for sub in range(max(k, 1)):
                B.launch_flags = (1 if sub > 0 else 0) | (2 if sub < k - 1 else 0) | _lib.LAUNCH_BODY_WRENCH | (sub << 8)
                if k > 0:
                    g["robot_prev_actions"][:] = g["robot_actions"]
                    g["robot_actions"][:] = a
                    self._robot_substep = sub
                    robot.step(g["robot_actions"])
                    if self._unknown_bodies:
                        self._unknown_wrench |= (F[:, self._unknown_bodies] != 0).any() | (T[:, self._unknown_bodies] != 0).any()
                    _lib.check(self._lib.agx_net_body_wrench(self.num_envs, _lib.C.byref(self._link_frames), _lib.dptr(F), _lib.dptr(T),
                                                             _lib.dptr(self._net_wrench), self._stream()), "agx_net_body_wrench")
                else:
                    self._net_wrench.zero_()
                _lib.check(self._lib.agx_env_step(P_wrench, B, self.num_envs, _lib.dptr(self._net_wrench), min(k, 1),
                                                  self.task_args, self._stream()), "agx_env_step")
            self._check_unknown_bodies()
        finally:
            B.launch_flags = 0
            self._robot_substep = 0
            self._in_external_simulate = False

    @roctx.ranged("EnvManager.step")
    def step(self, actions, env_actions=None):
        """env_actions: [N, num_assets, 6] obstacle twists (world-frame linear + angular velocity), the
        reference's dynamic-environment interface (env_manager.py:399-416, obstacle_manager.py:40-44)."""
        self._require_device()
        self._new_call()
        g = self.global_tensor_dict
        k = self.num_physics_steps()
        if env_actions is not None:
            if g["env_actions"] is None:  # the first tensor handed in becomes THE buffer, like the reference (:406-411)
                g["env_actions"], g["prev_env_actions"] = env_actions, env_actions
                self.env_actions, self.prev_env_actions = g["env_actions"], g["prev_env_actions"]
            self.prev_env_actions[:] = self.env_actions
            self.env_actions[:] = env_actions
            # the obstacles move first, then the robots fly k sub-steps against their new poses
            self.asset_manager.apply_env_actions(self, self.env_actions, k)
        # new env step: switch to the reset flag the previous step's reset kernel cleared
        self._parity ^= 1
        self._buffers.flag_parity = self._parity
        # counter word of the per-step device RNG streams (disturbance, observation / LiDAR noise, IMU)
        self._buffers.step_counter = self.step_counter & 0x7FFFFFFF
        if self._push is not None:
            self._advance_push()
        self.simulate(actions, env_actions, k)
        if self._push is not None:
            self._buffers.push_wait_seq = 0  # the env-step kernel has waited for the row slot; the kernels behind it need not
        self.step_counter += 1

    def compute_observations(self):
        """env_manager.py:358-362: collision_tensor |= contact.  The fused step (step / simulate) has already accumulated the flag
        over its sub-step positions; this is the stand-alone form for callers that drive simulate() and compute_observations()
        themselves (same predicate on the current position: calling it after a fused step changes nothing)."""
        self._require_device()
        _lib.check(self._lib.agx_collide_spheres_boxes(self._params, self._buffers, self.num_envs, self._stream()), "agx_collide_spheres_boxes")

    @roctx.ranged("EnvManager.post_reward_calculation_step")
    def post_reward_calculation_step(self):
        envs_to_reset = self.reset_terminated_and_truncated_envs()
        self.render(render_components="sensors")
        return envs_to_reset

    @roctx.ranged("EnvManager.render")
    def render(self, render_components="sensors"):
        if render_components == "sensors":
            self.render_sensors()

    def render_sensors(self):
        self.robot_manager.capture_sensors()

    def render_viewer(self):
        pass

    def get_obs(self):
        return self.global_tensor_dict

    def delete_env(self):
        """The reference's tasks call this on close() although its EnvManager lacks it
        (position_setpoint_task.py:132-133); provided here."""
        self._buffers = None
