"""PositionSetpointTask with the reference's API and step ordering
(aerial_gym/task/position_setpoint_task/position_setpoint_task.py:21-203); reward, crash,
truncation and observation packing run in agx_reward_position / agx_obs_position."""
import numpy as np
import torch

from .. import _lib
from ..sim.sim_builder import SimBuilder
from ..tensors import aos_view, soa
from ..utils.logging import CustomLogger
from ..utils.spaces import Box, Dict
from ..utils import roctx
from .base_task import BaseTask

logger = CustomLogger("position_setpoint_task")


class PositionSetpointTask(BaseTask):
    def __init__(self, task_config, seed=None, num_envs=None, headless=None, device=None, use_warp=None):
        for name, val in (("seed", seed), ("num_envs", num_envs), ("headless", headless), ("device", device),
                          ("use_warp", use_warp)):
            if val is not None:
                setattr(task_config, name, val)
        super().__init__(task_config)
        cfg = self.task_config
        self.device = cfg.device
        # args={"lean_step": True} (opt-in, effective above 65 536 envs: EnvManager._enable_lean_step) lets the fused step stop
        # maintaining the tensors that exist only to be looked at through the dict.  Never implied by num_envs: the dict behaves
        # the same at every batch size unless the caller asks otherwise.
        args = dict(cfg.args) if isinstance(cfg.args, dict) else cfg.args
        self.sim_env = SimBuilder().build_env(
            sim_name=cfg.sim_name, env_name=cfg.env_name, robot_name=cfg.robot_name,
            controller_name=cfg.controller_name, args=args, device=self.device, num_envs=cfg.num_envs,
            use_warp=cfg.use_warp, headless=cfg.headless,
        )
        N, dev = self.sim_env.num_envs, self.device
        self.num_envs = N
        self.sim_env.rows_written_twice_per_step = bool(cfg.return_state_before_reset)  # see sharding.StepGather
        self.actions = torch.zeros((N, cfg.action_space_dim), device=dev)
        self.prev_actions = torch.zeros_like(self.actions)
        self.counter = 0
        self.target_soa = soa(3, N, dev)
        self.target_position = aos_view(self.target_soa)
        self.obs_dict = self.sim_env.get_obs()
        self.obs_dict["num_obstacles_in_env"] = 1
        self.terminations = self.obs_dict["crashes"]
        self.truncations = self.obs_dict["truncations"]
        self.rewards = torch.zeros(N, device=dev)
        self.observation_space = Dict({"observations": Box(low=-1.0, high=1.0, shape=(13,), dtype=np.float32)})
        self.action_space = Box(low=-1.0, high=1.0, shape=(cfg.action_space_dim,), dtype=np.float32)
        self.task_obs = {
            "observations": torch.zeros((N, cfg.observation_space_dim), device=dev),
            "priviliged_obs": torch.zeros((N, cfg.privileged_observation_space_dim), device=dev),
            "collisions": torch.zeros((N, 1), device=dev),
            "rewards": torch.zeros((N, 1), device=dev),
        }
        self.infos = {}
        self._plan = None
        self._fuse_with_env()

    def _fuse_with_env(self):
        """Ask the EnvManager to evaluate this task's reward / crash / truncation / reset set in the
        env-step launch and its observation in the reset launch (2 launches per task.step())."""
        env = self.sim_env
        if env._buffers is None:
            return
        T = _lib.AgxTaskArgs()
        T.kind = _lib.TASK_POSITION
        T.episode_len = int(self.task_config.episode_len_steps)
        T.reset_on_collision = int(env.cfg.env.reset_on_collision)
        T.target = _lib.dptr(self.target_soa)
        T.reward = _lib.dptr(self.rewards)
        env.task_args = T
        env.post_obs = (_lib.dptr(self.target_soa), _lib.dptr(self.task_obs["observations"]))
        self._plan = self._strict = None
        e = env.cfg.env
        plain = (env.scene.num_assets == 0 and env.robot_manager.warp_sensor is None and env.robot_manager.imu_sensor is None
                  # host-evaluated controller / robot classes run between launches: the general path dispatches them
                  and not getattr(env.robot_manager.robot, "external_controller", False)
                  and not getattr(env.robot_manager.robot, "external_robot", False)
                  and not env.robot_manager.robot.cfg.disturbance.enable_disturbance
                  and e.num_physics_steps_per_env_step_std == 0 and not self.task_config.return_state_before_reset)
        draws = env.strict_draw_plan() if (plain and env.strict_rng) else None
        if plain and (not env.strict_rng or draws is not None):
            # whole task.step() = one host call launching two kernels (agx_position_task_step)
            import ctypes as C

            plan = _lib.AgxPositionStepPlan()
            plan.params, plan.buf = C.pointer(env._params), C.pointer(env._buffers)
            plan.task, plan.reset = C.pointer(T), C.pointer(env._reset_args)
            plan.target, plan.obs = env.post_obs
            plan.num_envs, plan.k_substeps = env.num_envs, int(e.num_physics_steps_per_env_step_mean)
            self._plan = plan
            self._plan_ref = C.byref(plan)       # (passed as it is every step: ctypes would build a new reference per call)
            self._plan_task = T                  # the struct plan.task points at (plan.task.contents builds a new object per access)
            self._plan_episode_len = T.episode_len
            self._plan_fn = env._lib.agx_position_task_step
            self._action_shape = (env.num_envs, env.num_robot_actions)
            self.task_obs["rewards"] = self.rewards
            self.task_obs["terminations"] = self.terminations
            self.task_obs["truncations"] = self.truncations
            if env.strict_rng:
                # reference-faithful RNG consumption, one host call per step: the two launches above with, between them, the
                # reset flag published into a mapped host word the call spins on (no stream synchronisation) and -- only on a
                # step with resets, like the reference's `if len(env_ids) > 0` -- the reset's seven rand_like draws as one launch
                # that reproduces torch's own numbers (csrc/agx_strict.hip); the generator's offset is moved by what they consume
                sp = _lib.AgxStrictStepPlan()
                if getattr(self, "_strict_word", None) is not None:  # (fused again after a re-bind)
                    env._lib.agx_host_word_destroy(self._strict_word)
                word = C.POINTER(C.c_uint32)()
                _lib.check(env._lib.agx_host_word_create(C.byref(word)), "agx_host_word_create")
                self._strict_word = word
                sp.plan, sp.host_word = C.pointer(plan), word
                sp.count = len(draws["tensors"])
                for j, t in enumerate(draws["tensors"]):
                    sp.out[j], sp.numel[j] = t.data_ptr(), t.numel()
                sp.sm_count, sp.max_threads_per_sm = draws["sm_count"], draws["max_threads_per_sm"]
                self._strict = sp
                self._strict_ref = C.byref(sp)
                self._strict_gen = draws["generator"]
                self._strict_drew, self._strict_after = C.c_int(0), C.c_uint64(0)
                self._strict_out = (C.byref(self._strict_drew), C.byref(self._strict_after))
                self._strict_fn = env._lib.agx_position_task_step_strict
                self._plan_fn = self._strict_step

    def _strict_step(self, plan_ref, actions_ptr, stream):
        """agx_position_task_step in the strict mode (same signature): generator state in, offset out"""
        sp, gen = self._strict, self._strict_gen
        self.sim_env.num_physics_steps()  # (the reference draws random.gauss(mean, std) every step, std = 0 included: same consumption)
        sp.seed, sp.offset = gen.initial_seed() & 0xFFFFFFFFFFFFFFFF, gen.get_offset()
        rc = self._strict_fn(self._strict_ref, actions_ptr, self._strict_out[0], self._strict_out[1], stream)
        if rc == 0 and self._strict_drew.value:
            gen.set_offset(self._strict_after.value)
        return rc

    def close(self):
        if getattr(self, "_strict_word", None) is not None:
            self.sim_env._lib.agx_host_word_destroy(self._strict_word)
            self._strict_word = None
        self.sim_env.delete_env()

    def reset(self):
        self.target_position[:, 0:3] = 0.0
        self.infos = {}
        self.sim_env.reset()
        return self.get_return_tuple()

    def reset_idx(self, env_ids):
        self.target_position[:, 0:3] = 0.0
        self.infos = {}
        self.sim_env.reset_idx(env_ids)

    def render(self):
        return None

    @roctx.ranged("PositionSetpointTask.step")
    def step(self, actions):
        self.counter += 1
        if (self._plan is not None and actions.dtype is torch.float32 and actions.is_contiguous() and actions.is_cuda
                and actions.shape == self._action_shape):  # anything else takes the general path, which raises like the reference
            # fast path: same two launches as the general path below, one host call.  Host time is 0.9 of this step at 8192 envs
            # (bench.py `host.share_of_step`): nothing here allocates or looks anything up twice.
            self.prev_actions = self.actions  # previous step's tensor (no copy; the reward does not read it)
            self.actions = actions
            env = self.sim_env
            env._stream_cache = None          # (EnvManager._new_call, inlined)
            env._derived_stale = True
            B = env._buffers
            B.step_counter = env.step_counter & 0x7FFFFFFF  # as EnvManager.step (RNG streams, step_signal)
            el = self.task_config.episode_len_steps
            if el != self._plan_episode_len:
                self._plan_task.episode_len = self._plan_episode_len = int(el)
            try:
                rc = self._plan_fn(self._plan_ref, actions.data_ptr(), _lib.current_stream(env.device))
            finally:
                env._parity = B.flag_parity  # the library toggles it first: stay in step on the error path too
            if rc != 0:
                _lib.check(rc, "agx_position_task_step")
            env._mask_fresh = env._obs_fresh = False
            env.step_counter += 1
            return (self.task_obs, self.rewards, self.terminations, self.truncations, self.infos)
        self.prev_actions = self.actions
        self.actions = actions
        env = self.sim_env
        if env.task_args is not None:
            env.task_args.episode_len = int(self.task_config.episode_len_steps)
        env.step(actions=self.actions)
        self.compute_rewards_and_crashes(self.obs_dict)
        if self.task_config.return_state_before_reset:
            return_tuple = self.get_return_tuple()
        env.post_reward_calculation_step()
        self.infos = {}
        if not self.task_config.return_state_before_reset:
            return_tuple = self.get_return_tuple()
        return return_tuple

    def compute_rewards_and_crashes(self, obs_dict):
        """compute_reward + `truncations = sim_steps > episode_len` (reference :205-229,:172-174)."""
        env = self.sim_env
        env._require_device()
        if env._reward_fresh:  # already produced by the fused epilogue of agx_env_step
            env._reward_fresh = False
            return self.rewards, self.terminations
        _lib.check(
            env._lib.agx_reward_position(env._buffers, env.num_envs, _lib.dptr(self.target_soa),
                                         int(self.task_config.episode_len_steps), int(env.cfg.env.reset_on_collision),
                                         _lib.dptr(self.rewards), env._stream()),
            "agx_reward_position",
        )
        env._mask_fresh = True  # the reward kernel wrote this step's reset set
        return self.rewards, self.terminations

    def get_return_tuple(self):
        self.process_obs_for_task()
        return (self.task_obs, self.rewards, self.terminations, self.truncations, self.infos)

    def process_obs_for_task(self):
        env = self.sim_env
        env._require_device()
        self.task_obs["rewards"] = self.rewards
        self.task_obs["terminations"] = self.terminations
        self.task_obs["truncations"] = self.truncations
        if env._obs_fresh:  # already written by agx_post_step_position
            env._obs_fresh = False
            return
        _lib.check(
            env._lib.agx_obs_position(env._buffers, env.num_envs, _lib.dptr(self.target_soa),
                                      _lib.dptr(self.task_obs["observations"]), env._stream()),
            "agx_obs_position",
        )
