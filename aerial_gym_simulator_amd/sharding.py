"""Multi-GPU env sharding: one process per GPU, envs split contiguously, ONE collective per
env step (RCCL all-gather over xGMI of the packed observation | reward | termination |
truncation rows).  The simulator itself needs no communication: envs never read each other
(SURVEY.md section 8e); the reference has no distributed path at all.

Payload per rank and step: N_local * (obs_dim + 3) * 4 B  (8192 envs, 13-D obs: 0.5 MB), i.e.
latency bound on a fully connected xGMI node -- hence a single fused buffer, not one collective
per tensor.
"""
import torch
import torch.distributed as dist


def shard_range(total_envs, rank, world_size):
    """Contiguous [lo, hi) env range of `rank`; remainders go to the first ranks."""
    base, rem = divmod(total_envs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def semantic_id_offset(rank, envs_per_rank, assets_per_env):
    """Start of this rank's slice of the reference's global segmentation counter
    (env_manager.py:147: 100 + one id per asset, counted across all envs)."""
    return rank * envs_per_rank * assets_per_env


class StepGather:
    """Packs (obs, reward, terminated, truncated) into one [N_local, obs_dim + 3] fp32 buffer and
    all-gathers it; `unpack` returns views of the gathered [world * N_local, ...] result."""

    def __init__(self, num_envs_local, obs_dim, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n, self.obs_dim = num_envs_local, obs_dim
        self.packed = torch.zeros(num_envs_local, obs_dim + 3, device=device)
        self.gathered = torch.zeros(self.world * num_envs_local, obs_dim + 3, device=device)

    def pack(self, obs, reward, terminated, truncated):
        p, d = self.packed, self.obs_dim
        p[:, :d] = obs
        p[:, d] = reward
        p[:, d + 1] = terminated
        p[:, d + 2] = truncated
        return p

    def gather(self, async_op=False):
        if self.world == 1:
            self.gathered.copy_(self.packed)
            return None
        return dist.all_gather_into_tensor(self.gathered, self.packed, group=self.group, async_op=async_op)

    def unpack(self):
        g, d = self.gathered, self.obs_dim
        return g[:, :d], g[:, d], g[:, d + 1] > 0.5, g[:, d + 2] > 0.5
