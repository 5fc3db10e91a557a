"""Golden sequence for the navigation task's glue (navigation_task.py:291-346 + :166-175, :234-270): the REAL
NavigationTask of the reference -- its own constructor, step(), reset_idx(), check_and_update_curriculum_level() --
running on a scripted stand-in for the simulator (SimBuilder().build_env returns it).  What the simulator would
produce (robot states, crash flags, images, which envs it resets) is scripted; everything the task derives from it
is the reference's: truncations, successes, timeouts, curriculum level / progress, target resampling.

    python oracle/gen_golden_nav_glue.py        (in the build container: needs /root/reference)
"""
import os
import sys

import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402  FIRST: it switches TorchScript off before torch is imported (cr_torch.py)

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shells  # noqa: E402

OUT = gg.OUT
N, T, H, W = 96, 60, 6, 8


class ScriptedSim:
    """EnvManager stand-in: the tensors a task reads, advanced from a pre-drawn script."""

    def __init__(self, g):
        self.num_envs = N
        self.g = g
        z = torch.zeros
        q = torch.randn(N, 4, generator=g)
        self.d = {
            "robot_position": z(N, 3), "robot_orientation": q / q.norm(dim=1, keepdim=True),
            "robot_vehicle_orientation": torch.tensor([0.0, 0.0, 0.0, 1.0]).repeat(N, 1), "robot_euler_angles": z(N, 3),
            "robot_body_linvel": z(N, 3), "robot_body_angvel": z(N, 3), "robot_actions": z(N, 4), "robot_prev_actions": z(N, 4),
            "crashes": z(N, dtype=torch.bool), "truncations": z(N, dtype=torch.bool),
            "env_bounds_min": -torch.rand(N, 3, generator=g) * 4.0 - 1.0, "env_bounds_max": torch.rand(N, 3, generator=g) * 4.0 + 1.0,
            "depth_range_pixels": torch.rand(N, 1, H, W, generator=g),
        }
        self.sim_steps = torch.zeros(N, dtype=torch.int32)
        self.log = []
        self.task = None

    def get_obs(self):
        return self.d

    def step(self, actions):
        g, d = self.g, self.d
        self.sim_steps += 1
        d["robot_prev_actions"][:] = d["robot_actions"]
        d["robot_actions"][:] = actions
        # a third of the envs sit within a metre or so of their target (success candidates), the rest wander
        # phases: mostly successful episodes (the level climbs), then mostly failures (it falls again)
        p_near = 0.97 if len(self.log) < 34 else 0.2
        near = torch.rand(N, generator=g) < p_near
        offset = torch.randn(N, 3, generator=g) * torch.where(near, 0.3, 3.0).unsqueeze(1)
        d["robot_position"][:] = self.task.target_position + offset
        d["robot_body_linvel"][:] = torch.randn(N, 3, generator=g)
        d["robot_body_angvel"][:] = torch.randn(N, 3, generator=g)
        d["robot_euler_angles"][:] = (torch.rand(N, 3, generator=g) - 0.5) * 6.0
        d["crashes"][:] = torch.rand(N, generator=g) < (0.004 if len(self.log) < 34 else 0.03)
        d["depth_range_pixels"][:] = torch.rand(N, 1, H, W, generator=g) * 1.2 - 0.1
        self.log.append({"position": d["robot_position"].clone(), "crashes": d["crashes"].clone(), "sim_steps": self.sim_steps.clone(),
                         "target_before": self.task.target_position.clone()})

    def post_reward_calculation_step(self):
        d = self.d
        ids = torch.nonzero(d["crashes"] | d["truncations"]).squeeze(-1)  # env_manager.py: terminated or truncated envs
        if len(ids) > 0:
            d["env_bounds_min"][ids] = -torch.rand(len(ids), 3, generator=self.g) * 4.0 - 1.0
            d["env_bounds_max"][ids] = torch.rand(len(ids), 3, generator=self.g) * 4.0 + 1.0
            self.sim_steps[ids] = 0
        self.log[-1].update(reset_ids=ids.clone(), bounds_min=d["env_bounds_min"].clone(), bounds_max=d["env_bounds_max"].clone())
        return ids

    def delete_env(self):
        pass


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_shells.install_task_shells()
    g = torch.Generator().manual_seed(2024)
    sim = ScriptedSim(g)

    class Builder:
        def build_env(self, **kw):
            return sim

    nt = ref_shells.ref("task.navigation_task.navigation_task")
    nt.SimBuilder = Builder
    from aerial_gym.config.task_config.navigation_task_config import task_config as ref_cfg

    ref_cfg.device = "cpu"  # action_transformation_function allocates on `task_config.device` by name

    class cfg(ref_cfg):
        episode_len_steps = 9
        num_envs = N
        device = "cpu"
        seed = 7
        use_warp = True
        headless = True

        class vae_config(ref_cfg.vae_config):
            use_vae = False

        class curriculum(ref_cfg.curriculum):
            check_after_log_instances = 64  # so that the level moves several times within the sequence

    cfg.reward_parameters = dict(ref_cfg.reward_parameters)
    task = nt.NavigationTask(cfg)
    sim.task = task
    torch.manual_seed(555)
    task.reset()
    out = {"initial_target": task.target_position.numpy().copy(), "initial_bounds_min": sim.d["env_bounds_min"].numpy().copy(),
           "initial_bounds_max": sim.d["env_bounds_max"].numpy().copy(), "initial_level": np.int64(task.curriculum_level),
           "curriculum": np.array([cfg.curriculum.min_level, cfg.curriculum.max_level, cfg.curriculum.check_after_log_instances,
                                   cfg.curriculum.increase_step, cfg.curriculum.decrease_step, cfg.curriculum.success_rate_for_increase,
                                   cfg.curriculum.success_rate_for_decrease], np.float64),
           "episode_len_steps": np.int64(cfg.episode_len_steps), "target_min_ratio": np.array(cfg.target_min_ratio, np.float32),
           "target_max_ratio": np.array(cfg.target_max_ratio, np.float32)}
    rows = {k: [] for k in ("position", "crashes", "sim_steps", "target_before", "truncations", "successes", "timeouts", "level",
                            "progress", "aggregates", "target_after", "reset_mask", "bounds_min", "bounds_max", "u_target", "rewards")}
    for t in range(T):
        actions = torch.rand(N, 4, generator=g) * 2 - 1
        # reset_idx draws rand_like [N, 3] from the global generator iff some env resets: replay it afterwards
        state = torch.get_rng_state()
        real_reset = task.reset_idx
        seen = {}

        def spy(env_ids, _real=real_reset):
            seen["rng"] = torch.get_rng_state()
            return _real(env_ids)

        task.reset_idx = spy
        # infos is cleared by reset_idx: read successes / timeouts through check_and_update_curriculum_level
        real_check = task.check_and_update_curriculum_level

        def check(successes, crashes, timeouts, _real=real_check):
            seen["successes"], seen["timeouts"] = successes.clone(), timeouts.clone()
            return _real(successes, crashes, timeouts)

        task.check_and_update_curriculum_level = check
        task.step(actions)
        task.reset_idx, task.check_and_update_curriculum_level = real_reset, real_check
        log = sim.log[-1]
        u = torch.zeros(N, 3)
        if "rng" in seen:
            after = torch.get_rng_state()
            torch.set_rng_state(seen["rng"])
            u = torch.rand(N, 3)
            torch.set_rng_state(after)
        mask = torch.zeros(N, dtype=torch.uint8)
        mask[log["reset_ids"]] = 1
        for k, v in (("position", log["position"]), ("crashes", log["crashes"]), ("sim_steps", log["sim_steps"]),
                     ("target_before", log["target_before"]), ("truncations", task.truncations.clone()),
                     ("successes", seen["successes"]), ("timeouts", seen["timeouts"]), ("level", torch.tensor(task.curriculum_level)),
                     ("progress", torch.tensor(float(task.curriculum_progress_fraction))),
                     ("aggregates", torch.tensor([int(task.success_aggregate), int(task.crashes_aggregate), int(task.timeouts_aggregate)])),
                     ("target_after", task.target_position.clone()), ("reset_mask", mask), ("bounds_min", log["bounds_min"]),
                     ("bounds_max", log["bounds_max"]), ("u_target", u), ("rewards", task.rewards.clone())):
            rows[k].append(v.numpy())
    for k, v in rows.items():
        out[k] = np.stack(v)
    np.savez_compressed(os.path.join(OUT, "navigation_glue.npz"), **out)
    print("navigation_glue: ok  levels", sorted(set(out["level"].tolist())), "successes", int(out["successes"].sum()), "timeouts",
          int(out["timeouts"].sum()), "crashes", int(out["crashes"].sum()), "resets", int(out["reset_mask"].sum()))


if __name__ == "__main__":
    main()
