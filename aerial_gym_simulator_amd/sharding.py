"""Multi-GPU env sharding: one process per GPU, envs split contiguously, ONE collective per
env step (RCCL all-gather over xGMI of the packed observation | reward | termination |
truncation rows).  The simulator itself needs no communication: envs never read each other
(SURVEY.md section 8e); the reference has no distributed path at all.

Payload per rank and step: N_local * (obs_dim + 3) * 4 B  (8192 envs, 13-D obs: 0.5 MB), i.e.
latency bound on a fully connected xGMI node -- hence
  * a single fused buffer, not one collective per tensor;
  * the send rows are written by the observation kernel itself (AgxEnvBuffers.step_rows), so
    the exchange adds no launch to the step;
  * rows and receive buffers are double buffered by step parity and the collective runs
    asynchronously on RCCL's own stream: the gather of step t overlaps the kernels of step t+1.
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
    """All-gathers [N_local, obs_dim + 3] fp32 rows (obs | reward | terminated | truncated) once
    per env step.

    With `env` (an EnvManager on a HIP device) the rows are produced by the observation kernels;
    without, the caller fills them with `pack()` (host-logic tests, custom tasks).

    exchange(parity, overlap) returns the gathered [world * N_local, obs_dim + 3] buffer (no
    device work besides the collective; `unpack` slices it):
      overlap=False  THIS step's rows of all ranks (stream-ordered);
      overlap=True   this step's collective is left in flight and the buffer of the PREVIOUS
                     step is returned (None on the first call) -- asynchronous samplers such
                     as the reference's Sample Factory recipe tolerate the one-step delay.
    """

    def __init__(self, num_envs_local, obs_dim, device, env=None, reward=None, group=None):
        self.group = group
        self.collective = dist.is_initialized()  # a world of one still goes through RCCL (bench debugging aid)
        self.world = dist.get_world_size(group) if self.collective else 1
        self.n, self.obs_dim = num_envs_local, obs_dim
        self.rows = torch.zeros(2, num_envs_local, obs_dim + 3, device=device)
        self.gathered = torch.zeros(2, self.world * num_envs_local, obs_dim + 3, device=device) if self.collective else self.rows
        self._work = [None, None]
        self._last = None
        if env is not None:
            env.bind_step_rows(self.rows, reward)

    def pack(self, parity, obs, reward, terminated, truncated):
        p, d = self.rows[parity], self.obs_dim
        p[:, :d] = obs
        p[:, d] = reward
        p[:, d + 1] = terminated
        p[:, d + 2] = truncated
        return p

    def exchange(self, parity, overlap=False):
        if self.collective:
            self._work[parity] = dist.all_gather_into_tensor(self.gathered[parity], self.rows[parity], group=self.group,
                                                             async_op=True)
        if not overlap:
            self.wait(parity)
            return self.gathered[parity]
        prev, self._last = self._last, parity
        if prev is None:
            return None
        # the next step writes rows[prev] again: its collective must have drained (stream wait, no host block)
        self.wait(prev)
        return self.gathered[prev]

    def wait(self, parity):
        w, self._work[parity] = self._work[parity], None
        if w is not None:
            w.wait()

    def flush(self):
        self.wait(0)
        self.wait(1)

    def unpack(self, buf):
        """(obs, reward, terminated, truncated) of a buffer returned by exchange()."""
        d = self.obs_dim
        return buf[:, :d], buf[:, d], buf[:, d + 1] > 0.5, buf[:, d + 2] > 0.5
