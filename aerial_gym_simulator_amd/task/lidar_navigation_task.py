"""LiDARNavigationTask with the reference's API and step ordering
(aerial_gym/task/lidar_navigation_task/lidar_navigation_task.py:19-470): the reference's flagship
training recipe (`magpie` + acceleration control + 48 x 120 dome LiDAR -> 337-D observation).

Per task.step(): agx_env_step (10 sub-steps) -> agx_reward_lidar_navigation (reward, crash /
truncation, reset set) -> resets -> sensor ray-cast (world-frame point cloud) -> agx_lidar_image_obs
(ranges, time to collision, 3 x 6 min-pool, noise, inverse) -> agx_obs_lidar_navigation.
Success / timeout bookkeeping and the curriculum stay as host logic like the reference."""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..tensors import aos_view, soa
from ..utils.logging import CustomLogger
from ..utils.spaces import Box, Dict
from .navigation_task import NavigationTask

logger = CustomLogger("lidar_navigation_task")


class LiDARNavigationTask(NavigationTask):
    def __init__(self, task_config, seed=None, num_envs=None, headless=None, device=None, use_warp=None):
        super().__init__(task_config, seed=seed, num_envs=num_envs, headless=headless, device=device, use_warp=use_warp)
        cfg, N, dev = self.task_config, self.num_envs, self.device
        px = self.obs_dict.get("depth_range_pixels")
        if px is None or px.dim() != 5 or px.shape[1] != 1:
            raise ValueError("LiDARNavigationTask needs one point-cloud LiDAR (depth_range_pixels [N, 1, H, W, 3])")
        sen = self.sim_env.robot_manager.warp_sensor.cfg
        if not sen.pointcloud_in_world_frame:
            raise ValueError("LiDARNavigationTask needs the point cloud in the world frame (rslidar_airy_config.py:17)")
        self._H, self._W = int(px.shape[2]), int(px.shape[3])
        self._ph, self._pw = cfg.lidar_pool
        self._cells = (self._H // self._ph) * (self._W // self._pw)
        if cfg.observation_space_dim != 17 + self._cells:
            raise ValueError(f"observation_space_dim {cfg.observation_space_dim} != 17 + {self._cells} pooled LiDAR cells")
        self.observation_space = Dict({"observations": Box(low=-np.inf, high=np.inf, shape=(cfg.observation_space_dim,), dtype=np.float32)})
        # the task's own (transformed) action history: the reward reads these, not robot_actions (:372-374, :493-494)
        self._action_ring = [torch.zeros(N, 4, device=dev), torch.zeros(N, 4, device=dev)]
        self._cur = 0
        self.time_to_collision = torch.zeros(N, device=dev)
        self.target_yaw = torch.zeros(N, device=dev)
        self.downsampled_lidar_data = torch.zeros(N, self._cells, device=dev)
        assert len(self._rp) == 22
        self._noise = None  # strict mode: the five noise tensors of add_noise_to_downsampled_lidar_data

    _zero_prev_actions_on_reset = True  # lidar_navigation_task.py:172

    def _target_yaw_ptr(self):
        return _lib.dptr(self.target_yaw)

    # the reward needs the task-level action history and the time to collision: it is its own launch
    def _fuse_with_env(self):
        self.sim_env.task_args = None

    @property
    def current_action(self):
        return self._action_ring[self._cur]

    @property
    def prev_action(self):
        return self._action_ring[self._cur ^ 1]

    def reset(self):
        self.sim_env.reset()
        self.reset_idx(torch.arange(self.sim_env.num_envs, device=self.device))
        self.sim_env.render(render_components="sensors")
        self.process_image_observation()
        return self.get_return_tuple()

    def reset_idx(self, env_ids):
        """Target + target yaw resample (:164-181): rand_like for all N / for the reset envs."""
        super().reset_idx(env_ids)
        rs = self.obs_dict["random_source"]
        mask_like = isinstance(env_ids, torch.Tensor) and env_ids.dtype in (torch.bool, torch.uint8)
        prev = self.obs_dict["robot_prev_actions"]
        if mask_like:
            m = env_ids.bool()
            yaw = (2.0 * torch.pi) * rs.rand(self.num_envs, tag="target_yaw") - torch.pi
            self.target_yaw[:] = torch.where(m, yaw, self.target_yaw)
            prev[:] = torch.where(m.unsqueeze(1), torch.zeros_like(prev), prev)
        else:
            n = len(env_ids)
            self.target_yaw[env_ids] = (2.0 * torch.pi) * rs.rand(n, tag="target_yaw") - torch.pi
            prev[env_ids] = 0.0

    def _flip_host_state(self):
        self._cur ^= 1  # prev_action <- current_action without a copy

    def _graph_key(self):
        return super()._graph_key() + (self._cur,)

    def _begin_step(self, transformed_action):
        self.current_action.copy_(transformed_action)

    def compute_rewards_and_crashes(self, obs_dict):
        env = self.sim_env
        env._require_device()
        p = _lib.dptr
        _lib.check(
            env._lib.agx_reward_lidar_navigation(env._buffers, env.num_envs, p(self.target_soa), p(self.target_yaw),
                                                 p(self.current_action), p(self.prev_action), p(self.time_to_collision), self._rp,
                                                 float(self.curriculum_progress_fraction), p(self.pos_err_soa),
                                                 p(self.prev_pos_err_soa), int(self.task_config.episode_len_steps),
                                                 int(env.cfg.env.reset_on_collision), p(self.rewards), env._stream()),
            "agx_reward_lidar_navigation",
        )
        env._mask_fresh = True  # the reward kernel wrote this step's reset set
        return self.rewards, self.terminations

    def _draw_lidar_noise(self):
        """strict mode: the draws of add_noise_to_downsampled_lidar_data (:281-310) in the reference's order
        (incl. its data-dependent draw count, i.e. one host sync)."""
        rs, N, dev = self.obs_dict["random_source"], self.num_envs, self.device
        oh, ow, r0 = self._H // self._ph, self._W // self._pw, int(self.task_config.lidar_low_noise_row0)
        noise_mask = rs.bernoulli(0.03, N, oh, ow, tag="lidar_noise_mask")
        k = int((noise_mask == 1).sum())
        noise_val = torch.zeros(N, oh, ow, device=dev)
        noise_val[noise_mask == 1] = (10.0 - 0.2) * rs.rand(k, tag="lidar_noise_val") + 0.2
        max_mask = rs.bernoulli(0.02, N, oh, ow, tag="lidar_max_mask")
        low_mask = torch.zeros(N, oh, ow, device=dev)
        low_val = torch.zeros(N, oh, ow, device=dev)
        low_mask[:, r0:] = rs.bernoulli(0.02, N, oh - r0, ow, tag="lidar_low_mask")
        low_val[:, r0:] = (1.0 - 0.2) * rs.rand(N, oh - r0, ow, tag="lidar_low_val") + 0.2
        self._noise = (noise_mask, noise_val, max_mask, low_mask, low_val)
        return [_lib.dptr(t) for t in self._noise]

    def process_image_observation(self):
        env = self.sim_env
        p = _lib.dptr
        noise = self._draw_lidar_noise() if env.strict_rng else [None] * 5
        _lib.check(
            env._lib.agx_lidar_image_obs(env._buffers, env.num_envs, self._H, self._W, self._ph, self._pw,
                                         int(self.task_config.lidar_low_noise_row0), p(self.obs_dict["depth_range_pixels"]),
                                         *noise, int(not env.strict_rng), p(self.time_to_collision),
                                         p(self.downsampled_lidar_data), env._stream()),
            "agx_lidar_image_obs",
        )

    def post_image_reward_addition(self):
        pass

    def process_obs_for_task(self):
        env = self.sim_env
        env._require_device()
        p = _lib.dptr
        uv = ue = None
        if env.strict_rng:
            rs = self.obs_dict["random_source"]
            uv, ue = p(rs.rand_into(self._u_vec, tag="obs_vec")), p(rs.rand_into(self._u_euler, tag="obs_euler"))
        _lib.check(
            env._lib.agx_obs_lidar_navigation(env._buffers, env.num_envs, p(self.target_soa), p(self.target_yaw), uv, ue,
                                              p(self.downsampled_lidar_data), self._cells, p(self.task_obs["observations"]),
                                              env._stream()),
            "agx_obs_lidar_navigation",
        )
