"""NavigationTask with the reference's API and step ordering
(aerial_gym/task/navigation_task/navigation_task.py:24-357).  Reward / truncation / reset set,
observation packing and the image min run in the HIP library; success / timeout bookkeeping
and the curriculum stay as (tiny) host logic like the reference.

The reference encodes the depth image with a pre-trained VAE (dense conv net, outside the
simulation hot path); with `vae_config.use_vae = False` (default here) the 64
latent slots carry an 8 x 8 min-pooled depth grid instead."""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..sim.sim_builder import SimBuilder
from ..tensors import aos_view, soa
from ..utils.logging import CustomLogger
from ..utils.spaces import Box, Dict
from ..utils import roctx
from .base_task import BaseTask

logger = CustomLogger("navigation_task")


class NavigationTask(BaseTask):
    _bookkeeping_fused = False  # (set by _fuse_with_env; subclasses that hand the env no task arguments leave it off)
    _bookkeeping_ptrs = None

    def __init__(self, task_config, seed=None, num_envs=None, headless=None, device=None, use_warp=None):
        for name, val in (("seed", seed), ("num_envs", num_envs), ("headless", headless), ("device", device),
                          ("use_warp", use_warp)):
            if val is not None:
                setattr(task_config, name, val)
        super().__init__(task_config)
        cfg = self.task_config
        self.device = cfg.device
        self._args = dict(cfg.args) if isinstance(cfg.args, dict) else {}  # a snapshot: the config object is shared between tasks
        self.sim_env = SimBuilder().build_env(
            sim_name=cfg.sim_name, env_name=cfg.env_name, robot_name=cfg.robot_name,
            controller_name=cfg.controller_name, args=cfg.args, device=self.device, num_envs=cfg.num_envs,
            use_warp=cfg.use_warp, headless=cfg.headless,
        )
        N, dev = self.sim_env.num_envs, self.device
        self.num_envs = N
        self.sim_env.rows_written_twice_per_step = bool(cfg.return_state_before_reset)  # see sharding.StepGather
        self.target_soa = soa(3, N, dev)
        self.target_position = aos_view(self.target_soa)
        self.target_min_ratio = torch.tensor(cfg.target_min_ratio, device=dev).expand(N, -1)
        self.target_max_ratio = torch.tensor(cfg.target_max_ratio, device=dev).expand(N, -1)
        self.success_aggregate = self.crashes_aggregate = self.timeouts_aggregate = 0
        self.pos_err_soa, self.prev_pos_err_soa = soa(3, N, dev), soa(3, N, dev)
        self.pos_error_vehicle_frame = aos_view(self.pos_err_soa)
        self.pos_error_vehicle_frame_prev = aos_view(self.prev_pos_err_soa)
        if cfg.vae_config.use_vae:
            raise NotImplementedError("the VAE image encoder is out of scope (SURVEY.md row 14); set use_vae = False")
        self.obs_dict = self.sim_env.get_obs()
        if "curriculum_level" not in self.obs_dict:
            self.curriculum_level = cfg.curriculum.min_level
            self.obs_dict["curriculum_level"] = self.curriculum_level
        else:
            self.curriculum_level = self.obs_dict["curriculum_level"]
        self.obs_dict["num_obstacles_in_env"] = self.curriculum_level
        self._update_progress()
        self.terminations = self.obs_dict["crashes"]
        self.truncations = self.obs_dict["truncations"]
        self.rewards = torch.zeros(N, device=dev)
        self.min_pixel_dist = torch.zeros(N, device=dev)
        self.observation_space = Dict(
            {"observations": Box(low=-1.0, high=1.0, shape=(cfg.observation_space_dim,), dtype=np.float32)}
        )
        self.action_space = Box(low=-1.0, high=1.0, shape=(4,), dtype=np.float32)
        self.action_transformation_function = cfg.action_transformation_function
        self._action_kind = getattr(cfg.action_transformation_function, "agx_kind", None)  # built-in: one launch (agx_action_transform)
        self._action_out = None
        self.task_obs = {"observations": torch.zeros((N, cfg.observation_space_dim), device=dev)}
        self.num_task_steps = 0
        self.infos = {}
        order = cfg.REWARD_PARAMETER_ORDER
        self._rp = (C.c_float * len(order))(*[float(cfg.reward_parameters[k]) for k in order])
        self._u_vec = torch.zeros(N, 3, device=dev)
        self._u_euler = torch.zeros(N, 3, device=dev)
        # sync-free mode: bookkeeping on the device; the host only reads the three counters every
        # `curriculum_check_every` steps (the reference syncs on them every step, navigation_task.py:237)
        default_every = 1 if self.sim_env.strict_rng else 16
        self.curriculum_check_every = int(cfg.args.get("curriculum_check_every", default_every)) if isinstance(cfg.args, dict) else default_every
        self._successes = torch.zeros(N, dtype=torch.bool, device=dev)
        self._timeouts = torch.zeros(N, dtype=torch.bool, device=dev)
        self._counters = torch.zeros(3, dtype=torch.int32, device=dev)
        self._graphs, self._graph_action = None, None  # see _graph_mode()
        self._graph_wanted = bool(self._arg("step_graph", False))
        self.sim_env.step_graph_mode = self._graph_wanted  # (sharding.StepGather: no kernel-side push inside a replayed graph)
        self._min_ratio = (C.c_float * 3)(*[float(v) for v in cfg.target_min_ratio])
        self._max_ratio = (C.c_float * 3)(*[float(v) for v in cfg.target_max_ratio])
        self._fuse_with_env()

    def _arg(self, key, default):
        """switches of this task come through task_config.args as it was at construction (the dict that also reaches the
        EnvManager), never through the process environment"""
        return self._args.get(key, default)

    def _fuse_with_env(self):
        env = self.sim_env
        if env._buffers is None:
            return
        T = _lib.AgxTaskArgs()
        T.kind = _lib.TASK_NAVIGATION
        T.episode_len = int(self.task_config.episode_len_steps)
        T.reset_on_collision = int(env.cfg.env.reset_on_collision)
        T.curriculum_progress = float(self.curriculum_progress_fraction)
        T.target, T.reward = _lib.dptr(self.target_soa), _lib.dptr(self.rewards)
        T.pos_err, T.prev_pos_err = _lib.dptr(self.pos_err_soa), _lib.dptr(self.prev_pos_err_soa)
        for i in range(18):
            T.rp[i] = self._rp[i]
        # sync-free mode: successes / timeouts / curriculum counters of the step come out of the same epilogue (the arithmetic of
        # agx_nav_bookkeeping on the launch's registers: one dispatch fewer per step).  args={"fused_bookkeeping": False}: the launch of its own.
        self._bookkeeping_fused = not env.strict_rng and bool(self._arg("fused_bookkeeping", True))
        if self._bookkeeping_fused:
            # (the pointers are handed over only around THIS task's own env.step, _device_step: an env.step() a caller issues on
            # task.sim_env by hand must not move the curriculum counters -- they moved only in _bookkeeping_device before round 5)
            self._bookkeeping_ptrs = (_lib.dptr(self._successes), _lib.dptr(self._timeouts), _lib.dptr(self._counters))
            T.success_radius = 1.0
        env.task_args = T

    def _update_progress(self):
        c = self.task_config.curriculum
        self.curriculum_progress_fraction = (self.curriculum_level - c.min_level) / (c.max_level - c.min_level)

    def close(self):
        self.sim_env.delete_env()

    def reset(self):
        self.sim_env.reset()
        self.reset_idx(torch.arange(self.sim_env.num_envs, device=self.device))
        self.sim_env.render(render_components="sensors")
        return self.get_return_tuple()

    def reset_idx(self, env_ids):
        """Target resample (navigation_task.py:166-175): rand_like for all N, then index."""
        rs = self.obs_dict["random_source"]
        ratio = (self.target_max_ratio - self.target_min_ratio) * rs.rand(self.num_envs, 3, tag="target") + self.target_min_ratio
        bmin, bmax = self.obs_dict["env_bounds_min"], self.obs_dict["env_bounds_max"]
        new = bmin + (bmax - bmin) * ratio
        if isinstance(env_ids, torch.Tensor) and env_ids.dtype in (torch.bool, torch.uint8):
            self.target_position[:] = torch.where(env_ids.bool().unsqueeze(1), new, self.target_position)
        else:
            self.target_position[env_ids] = new[env_ids]
        self.infos = {}

    def render(self):
        return self.sim_env.render()

    def check_and_update_curriculum_level(self, successes, crashes, timeouts):
        self.success_aggregate += torch.sum(successes)
        self.crashes_aggregate += torch.sum(crashes)
        self.timeouts_aggregate += torch.sum(timeouts)
        if self.num_task_steps % self.curriculum_check_every != 0:
            return
        self._curriculum_decision(int(self.success_aggregate), int(self.crashes_aggregate), int(self.timeouts_aggregate))  # host sync (:237)

    def _curriculum_decision(self, n_success, n_crash, n_timeout):
        """navigation_task.py:229-270 on host integers; returns True when the aggregates were consumed."""
        instances = n_success + n_crash + n_timeout
        c = self.task_config.curriculum
        if instances < c.check_after_log_instances:
            return False
        success_rate = n_success / instances
        if success_rate > c.success_rate_for_increase:
            self.curriculum_level += c.increase_step
        elif success_rate < c.success_rate_for_decrease:
            self.curriculum_level -= c.decrease_step
        self.curriculum_level = min(max(self.curriculum_level, c.min_level), c.max_level)
        self.obs_dict["curriculum_level"] = self.curriculum_level
        self.obs_dict["num_obstacles_in_env"] = self.curriculum_level
        self._update_progress()
        logger.warning(f"Curriculum Level: {self.curriculum_level}, success rate {success_rate:.3f}")
        self.success_aggregate = self.crashes_aggregate = self.timeouts_aggregate = 0
        return True

    def _bookkeeping_device(self):
        """successes / timeouts / crashes of this step (navigation_task.py:311-326) and the curriculum counters"""
        env = self.sim_env
        if env.strict_rng:  # reference-faithful: torch reductions and a host sync on the aggregates
            near = torch.norm(self.target_position - self.obs_dict["robot_position"], dim=1) < 1.0
            crashed = self.terminations
            successes = self.truncations & near & ~crashed
            timeouts = self.truncations & ~successes & ~crashed
            self.infos["successes"], self.infos["timeouts"], self.infos["crashes"] = successes, timeouts, crashed
            self.check_and_update_curriculum_level(successes, crashed, timeouts)
            return
        p = _lib.dptr
        if not (getattr(self, "_bookkeeping_fused", False) and env.task_args is not None):
            _lib.check(env._lib.agx_nav_bookkeeping(env._buffers, env.num_envs, p(self.target_soa), 1.0, p(self._successes),
                                                    p(self._timeouts), p(self._counters), env._stream()), "agx_nav_bookkeeping")
        self.infos["successes"], self.infos["timeouts"], self.infos["crashes"] = self._successes, self._timeouts, self.terminations

    def _bookkeeping_host(self, in_step=False):
        """the curriculum (navigation_task.py:229-270): the only host synchronisation of the task, every
        `curriculum_check_every` steps.  Eager stepping runs it where the reference does -- after this step's bookkeeping and
        BEFORE post_reward_calculation_step (navigation_task.py:327-331), so that the resets of the check step already see the
        new level / num_obstacles_in_env (`in_step`).  A step that is being captured into, or replayed from, a hipGraph cannot
        synchronise in the middle: there the check runs after the step and the level applies from the next step on (a
        one-step delay of every level change, INTEGRATION.md "known deviations")."""
        if self.sim_env.strict_rng:
            return
        if in_step:
            if self._in_graph_step or self.num_task_steps % self.curriculum_check_every != 0:
                return
            self._curriculum_checked_in_step = True
        else:
            if self._curriculum_checked_in_step:
                self._curriculum_checked_in_step = False
                return
            if (self.num_task_steps - 1) % self.curriculum_check_every != 0:
                return
        ns, nc, nt = self._counters.tolist()
        if self._curriculum_decision(ns, nc, nt):
            self._counters.zero_()

    _in_graph_step = False
    _curriculum_checked_in_step = False

    def _reset_targets(self, reset_envs):
        """reset_idx for the envs that were just reset"""
        env = self.sim_env
        if env.strict_rng:
            if len(reset_envs) > 0:
                self.reset_idx(reset_envs.indices)
            return
        if env._targets_reset_fused:  # done by the launch that reset the robots (agx_nav_robot_side)
            env._targets_reset_fused = False
            self.infos = {}
            return
        _lib.check(env._lib.agx_nav_target_reset(env._buffers, env.num_envs, env.num_robot_actions, self._min_ratio, self._max_ratio, None,
                                                 _lib.dptr(self.target_soa), self._target_yaw_ptr(), int(self._zero_prev_actions_on_reset),
                                                 env._stream()), "agx_nav_target_reset")
        self.infos = {}

    _zero_prev_actions_on_reset = False
    _fused_side = None  # False: not applicable; else the AgxNavRobotSideArgs handed to the EnvManager

    def _setup_fused_robot_side(self):
        """Sync-free mode: the robot reset, the sensor mounts and the target of the envs that reset and the pose of every sensor
        as ONE launch per step (agx_nav_robot_side) instead of four.  The stand-alone entry points stay
        what explicit reset_idx() / render() calls use; both run the same device functions."""
        env = self.sim_env
        rm = env.robot_manager
        wanted = bool(self._arg("fused_robot_side", True))
        if not wanted or env.strict_rng or env._buffers is None or rm.imu_sensor is not None or env.post_obs is not None:
            self._fused_side = False
            return
        A = _lib.AgxNavRobotSideArgs()
        sensor = rm.warp_sensor
        if sensor is not None:
            A.num_sensors = int(sensor.num_sensors)
            A.randomize_mount = int(bool(sensor.cfg.randomize_placement))
            for c in range(3):
                A.mount_t_min[c], A.mount_t_max[c] = sensor._min_t[c], sensor._max_t[c]
                A.mount_r_min[c], A.mount_r_max[c] = sensor._min_r[c], sensor._max_r[c]
            for c in range(4):
                A.frame_quat[c] = sensor.frame_quat[c]
            A.local_pos, A.local_quat = _lib.dptr(sensor.sensor_local_position), _lib.dptr(sensor.sensor_local_orientation)
            A.sensor_pos, A.sensor_quat = _lib.dptr(sensor.sensor_position), _lib.dptr(sensor.sensor_orientation)
        A.reset_target = 1
        A.num_actions = int(env.num_robot_actions)
        A.zero_prev_actions = int(self._zero_prev_actions_on_reset)
        for c in range(3):
            A.target_ratio_min[c], A.target_ratio_max[c] = self._min_ratio[c], self._max_ratio[c]
        A.target = _lib.dptr(self.target_soa)
        A.target_yaw = self._target_yaw_ptr()
        self._fused_side = A
        env.enable_fused_robot_side(A)

    def _target_yaw_ptr(self):
        return None

    # ------------------------------------------------------------------ stepping
    @roctx.ranged("NavigationTask.step")
    def step(self, actions):
        transformed_action = self._transform_action(actions)
        self._in_graph_step = bool(self._graph_mode()) or torch.cuda.is_current_stream_capturing()
        if self._graph_mode():
            return self._step_replayed(transformed_action)
        self._flip_host_state()
        ret = self._device_step(transformed_action)
        self._finish_step_host()
        return ret

    def _transform_action(self, actions):
        kind, env = self._action_kind, self.sim_env
        if (kind is None or env._buffers is None or not actions.is_cuda or actions.dtype is not torch.float32 or not actions.is_contiguous()
                or actions.shape != (self.num_envs, 4)):
            return self.action_transformation_function(actions)
        env._new_call()
        out = torch.empty((self.num_envs, kind[1]), device=actions.device)  # a fresh tensor per step, like the torch form (callers keep references)
        _lib.check(env._lib.agx_action_transform(kind[0], self.num_envs, _lib.dptr(actions), _lib.dptr(out), env._stream()),
                   "agx_action_transform")
        return out

    def _flip_host_state(self):
        """host-side state that alternates per step and decides which buffers the step's kernels are handed (none here;
        the LiDAR task's action ring)"""

    def _device_step(self, transformed_action):
        """Everything task.step() puts on the device, in the reference's order (navigation_task.py:291-349), with no
        host synchronisation in the sync-free mode: this is the region a hipGraph captures for small batches."""
        env = self.sim_env
        self._begin_step(transformed_action)
        if env.task_args is not None:
            env.task_args.curriculum_progress = float(self.curriculum_progress_fraction)
            env.task_args.episode_len = int(self.task_config.episode_len_steps)
        T = env.task_args
        fused = self._bookkeeping_fused and T is not None
        if fused:
            T.successes, T.timeouts, T.counters = self._bookkeeping_ptrs
        try:
            env.step(actions=transformed_action)
        finally:
            if fused:
                T.successes = T.timeouts = T.counters = None
        self.compute_rewards_and_crashes(self.obs_dict)
        if self.task_config.return_state_before_reset:
            return_tuple = self.get_return_tuple()
        self._bookkeeping_device()
        self._bookkeeping_host(in_step=True)
        if self._fused_side is None:
            self._setup_fused_robot_side()
        reset_envs = env.post_reward_calculation_step()
        self._reset_targets(reset_envs)
        self.process_image_observation()
        # one sensor and the observation taken from the same frame: the observation kernel's sweep over the image also
        # produces post_image_reward_addition's minimum (one read of the image instead of two)
        self._min_pixel_in_obs = self._image_min_fusable()
        if not self._min_pixel_in_obs:
            self.post_image_reward_addition()
        if not self.task_config.return_state_before_reset:
            return_tuple = self.get_return_tuple()
        self._min_pixel_in_obs = False
        if env._step_counter_dev is not None:  # replayed steps read the step index from device memory
            _lib.check(env._lib.agx_step_counter_advance(env._buffers, env._stream()), "agx_step_counter_advance")
        return return_tuple

    def _begin_step(self, transformed_action):
        pass

    def _finish_step_host(self):
        self.num_task_steps += 1
        self._bookkeeping_host()

    # ---- opt-in (args={"step_graph": True}): the step as a replayed hipGraph -----------------------
    # In the sync-free mode the step has no host synchronisation and fixed buffers, so it can be captured once per
    # (reset-flag parity, curriculum level[, action-ring slot]) and replayed: one launch per step instead of ~15.  What
    # changes from step to step travels through memory: the action (a static buffer), the step index
    # (AgxEnvBuffers.step_counter_dev).  Host-side curriculum logic runs between replays, exactly as in eager mode.
    # Measured (profiles/r02_small_batch.txt): NO gain at 256 .. 1024 envs (319 vs 324 us per step at 256 envs) -- the step
    # is bound by the GPU-side chain of small dependent kernels, not by their launches -- hence off by default; a caller
    # who captures policy + step in one graph gets a capture-safe step either way.
    GRAPH_WARMUP_STEPS = 3

    def _graph_mode(self):
        if self._graphs is None:  # decided once
            env = self.sim_env
            sensor = env.robot_manager.warp_sensor
            ok = (self._graph_wanted and env._buffers is not None and not env.strict_rng and env.robot_manager.imu_sensor is None
                  and (sensor is None or not sensor.cfg.sensor_noise.enable_sensor_noise)  # torch RNG ops in the step
                  and env.cfg.env.num_physics_steps_per_env_step_std == 0)
            self._graphs = {} if ok else False
        if self._graphs is False:
            return False
        return self.num_task_steps >= self.GRAPH_WARMUP_STEPS  # the first steps run eagerly (lazy initialisation)

    def _graph_key(self):
        return (self.sim_env._parity ^ 1, self.curriculum_level)

    def _step_replayed(self, transformed_action):
        env = self.sim_env
        if self._graph_action is None:
            self._graph_action = torch.zeros_like(transformed_action)
            env.enable_device_step_counter()
        self._graph_action.copy_(transformed_action)
        self._flip_host_state()
        key = self._graph_key()
        entry = self._graphs.get(key)
        if entry is None:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):  # the capture performs the step's HOST effects (parity, counters), no device work
                    ret = self._device_step(self._graph_action)
                self._graphs[key] = (graph, ret)
                graph.replay()
            except Exception as e:  # noqa: BLE001  (capture is an optimisation: fall back to eager stepping for good)
                logger.warning(f"hipGraph capture of the step failed ({type(e).__name__}: {e}); stepping eagerly")
                self._graphs = False
                ret = self._device_step(transformed_action)
        else:
            graph, ret = entry
            env._new_call()
            env._parity ^= 1  # what EnvManager.step does on the host (the kernels' copies are frozen in the graph)
            env._buffers.flag_parity = env._parity
            env.step_counter += 1
            env._reward_fresh = env._obs_fresh = env._mask_fresh = False
            graph.replay()
        self._finish_step_host()
        return ret

    def process_image_observation(self):
        pass  # the min-pooled latents are written by agx_obs_navigation

    _min_pixel_in_obs = False

    def _image_min_fusable(self):
        px = self.obs_dict.get("depth_range_pixels")
        return (px is not None and px.dim() == 4 and px.shape[1] == 1 and self.task_config.observation_space_dim > 17
                and not self.task_config.return_state_before_reset)

    def post_image_reward_addition(self):
        """navigation_task.py:351-357.  `rewards[terminations < 0] += ...` never selects anything
        for a bool tensor, so only min_pixel_dist is produced (reference quirk, kept)."""
        env, px = self.sim_env, self.obs_dict.get("depth_range_pixels")
        if px is None or px.dim() != 4:
            return
        ppe = px.shape[1] * px.shape[2] * px.shape[3]
        _lib.check(env._lib.agx_image_min(env.num_envs, ppe, _lib.dptr(px), _lib.dptr(self.min_pixel_dist), env._stream()),
                   "agx_image_min")

    def compute_rewards_and_crashes(self, obs_dict):
        env = self.sim_env
        env._require_device()
        if env._reward_fresh:  # produced by the fused epilogue of agx_env_step
            env._reward_fresh = False
            return self.rewards, self.terminations
        _lib.check(
            env._lib.agx_reward_navigation(env._buffers, env.num_envs, _lib.dptr(self.target_soa), self._rp,
                                           float(self.curriculum_progress_fraction), _lib.dptr(self.pos_err_soa),
                                           _lib.dptr(self.prev_pos_err_soa), int(self.task_config.episode_len_steps),
                                           int(env.cfg.env.reset_on_collision), _lib.dptr(self.rewards), env._stream()),
            "agx_reward_navigation",
        )
        env._mask_fresh = True  # the reward kernel wrote this step's reset set
        return self.rewards, self.terminations

    def get_return_tuple(self):
        self.process_obs_for_task()
        return (self.task_obs, self.rewards, self.terminations, self.truncations, self.infos)

    def process_obs_for_task(self):
        env = self.sim_env
        env._require_device()
        uv = ue = None  # sync-free: the kernel draws from the device generator
        if env.strict_rng:
            rs = self.obs_dict["random_source"]
            uv, ue = _lib.dptr(rs.rand_into(self._u_vec, tag="obs_vec")), _lib.dptr(rs.rand_into(self._u_euler, tag="obs_euler"))
        px = self.obs_dict.get("depth_range_pixels")
        use_px = px is not None and px.dim() == 4 and self.task_config.observation_space_dim > 17
        S, H, W = (px.shape[1], px.shape[2], px.shape[3]) if use_px else (0, 0, 0)
        _lib.check(
            env._lib.agx_obs_navigation(env._buffers, env.num_envs, _lib.dptr(self.target_soa), uv, ue,
                                        _lib.dptr(px) if use_px else None, S, H, W, 8, 8,
                                        int(self.task_config.observation_space_dim),
                                        _lib.dptr(self.task_obs["observations"]),
                                        _lib.dptr(self.min_pixel_dist) if (use_px and self._min_pixel_in_obs) else None,
                                        env._stream()),
            "agx_obs_navigation",
        )
