"""Test helper: the two ways the reference's trainers consume a task, restated from their call sites so that the GPU box
(no /root/reference) can exercise them.  When the reference tree is present the tests load ITS classes instead.

* rl_games (rl_training/rl_games/runner.py:26-80): a wrapper whose reset() returns obs["observations"] and whose
  step() returns (obs["observations"], rewards, terminated | truncated as the flags' dtype, infos); the vec-env reads
  task_config.action_space_dim / observation_space_dim.
* cleanrl (rl_training/cleanrl/ppo_continuous_action.py:239-276): per-env episode return / length bookkeeping with
  dones = where(terminated | truncated, 1, 0), written INTO the task's infos dict ("r", "l")."""
import torch


class Forwarding:
    """what gym.Wrapper does for these scripts: keep `env`, forward unknown attributes to it"""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(action)


class ObsExtractor(Forwarding):
    def reset(self, **kwargs):
        observations, *_ = self.env.reset(**kwargs)
        return observations["observations"]

    def step(self, action):
        observations, rewards, terminated, truncated, infos = self.env.step(action)
        dones = torch.where(terminated | truncated, torch.ones_like(terminated), torch.zeros_like(terminated))
        return observations["observations"], rewards, dones, infos


class EpisodeStatistics(Forwarding):
    def __init__(self, env, device):
        super().__init__(env)
        self.num_envs, self.device = getattr(env, "num_envs", 1), device

    def reset(self, **kwargs):
        out = self.env.reset(**kwargs)
        z = lambda dt: torch.zeros(self.num_envs, dtype=dt, device=self.device)  # noqa: E731
        self.episode_returns, self.episode_lengths = z(torch.float32), z(torch.int32)
        self.returned_episode_returns, self.returned_episode_lengths = z(torch.float32), z(torch.int32)
        return out

    def step(self, action):
        observations, rewards, terminations, truncations, infos = self.env.step(action)
        self.episode_returns += rewards
        self.episode_lengths += 1
        self.returned_episode_returns[:] = self.episode_returns
        self.returned_episode_lengths[:] = self.episode_lengths
        dones = torch.where(terminations | truncations, 1, 0).to(self.device)
        self.episode_returns *= 1 - dones
        self.episode_lengths *= 1 - dones
        infos["r"], infos["l"] = self.returned_episode_returns, self.returned_episode_lengths
        return observations, rewards, dones, infos
