"""PPO on the position set-point task, on the GPU end to end: a caller of the Task API in the shape of the reference's
trainers (rl_training/cleanrl/ppo_continuous_action.py:239-277 builds the same objects: task_registry.make_task, a statistics
wrapper around reset() / step(), a 256-256 tanh actor / critic).  The trainers themselves are out of this repo's scope
(SURVEY section 8); this script exists to show that the simulator TRAINS a policy and how long that takes on an MI355X --
the reference's documentation quotes "under a minute on an RTX 3090" for this task (docs/6_rl_training.md:134-140, rl_games).

    python examples/train_ppo_position.py --num_envs 8192 --seconds 60 --out profiles/r03_ppo_position_training.json

Hyper-parameters default to the reference script's (learning rate 2.6e-3, 32-step rollouts, 2 minibatches, 4 epochs,
gamma 0.99, lambda 0.95, value coefficient 2, gradient clip 1).  Everything -- policy, rollout storage, GAE, updates -- stays
on the device; the host reads four scalars per update for the log.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def layer(i, o, std=2 ** 0.5):
    m = nn.Linear(i, o)
    nn.init.orthogonal_(m.weight, std)
    nn.init.zeros_(m.bias)
    return m


class ActorCritic(nn.Module):
    def __init__(self, obs_dim, act_dim):
        super().__init__()
        self.critic = nn.Sequential(layer(obs_dim, 256), nn.Tanh(), layer(256, 256), nn.Tanh(), layer(256, 1, 1.0))
        self.actor = nn.Sequential(layer(obs_dim, 256), nn.Tanh(), layer(256, 256), nn.Tanh(), layer(256, act_dim, 0.01))
        self.logstd = nn.Parameter(torch.zeros(1, act_dim))

    def dist(self, obs):
        mean = self.actor(obs)
        return torch.distributions.Normal(mean, self.logstd.expand_as(mean).exp())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_envs", type=int, default=8192)
    ap.add_argument("--seconds", type=float, default=60.0, help="wall-clock training budget")
    ap.add_argument("--max_updates", type=int, default=100000)
    ap.add_argument("--num_steps", type=int, default=32)
    ap.add_argument("--lr", type=float, default=0.0026)
    ap.add_argument("--gamma", type=float, default=0.99)
    ap.add_argument("--lam", type=float, default=0.95)
    ap.add_argument("--minibatches", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--clip", type=float, default=0.2)
    ap.add_argument("--vf_coef", type=float, default=2.0)
    ap.add_argument("--max_grad_norm", type=float, default=1.0)
    ap.add_argument("--controller", default=None, help="default: the task config's (lee_attitude_control, like the reference)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    dev = "cuda:0"
    torch.manual_seed(args.seed)
    cfg.device = dev
    if args.controller:
        cfg.controller_name = args.controller
    task = task_registry.make_task("position_setpoint_task", seed=args.seed, num_envs=args.num_envs, headless=True)
    N, T = args.num_envs, args.num_steps
    obs_dim, act_dim = cfg.observation_space_dim, cfg.action_space_dim
    agent = ActorCritic(obs_dim, act_dim).to(dev)
    opt = torch.optim.Adam(agent.parameters(), lr=args.lr, eps=1e-5)
    obs_b = torch.zeros(T, N, obs_dim, device=dev)
    act_b = torch.zeros(T, N, act_dim, device=dev)
    logp_b, rew_b, done_b, val_b = (torch.zeros(T, N, device=dev) for _ in range(4))
    ep_ret, ep_len = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    fin_ret_sum, fin_len_sum, fin_n = (torch.zeros((), device=dev) for _ in range(3))

    next_obs = task.reset()[0]["observations"].clone()
    next_done = torch.zeros(N, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    log, env_steps, sim_time = [], 0, 0.0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for update in range(1, args.max_updates + 1):
        ev0.record()
        for t in range(T):
            obs_b[t], done_b[t] = next_obs, next_done
            with torch.no_grad():
                d = agent.dist(next_obs)
                a = d.sample()
                logp_b[t] = d.log_prob(a).sum(1)
                val_b[t] = agent.critic(next_obs).flatten()
            act_b[t] = a
            o, r, term, trunc, _ = task.step(a)
            next_obs = o["observations"].clone()  # the task reuses its observation buffer
            rew_b[t] = r
            next_done = (term | trunc).float()
            ep_ret += r
            ep_len += 1
            fin_ret_sum += (ep_ret * next_done).sum()
            fin_len_sum += (ep_len * next_done).sum()
            fin_n += next_done.sum()
            ep_ret *= 1 - next_done
            ep_len *= 1 - next_done
        ev1.record()
        env_steps += N * T
        with torch.no_grad():  # GAE
            adv = torch.zeros_like(rew_b)
            last = torch.zeros(N, device=dev)
            next_val = agent.critic(next_obs).flatten()
            for t in reversed(range(T)):
                nonterm = 1.0 - (next_done if t == T - 1 else done_b[t + 1])
                nv = next_val if t == T - 1 else val_b[t + 1]
                delta = rew_b[t] + args.gamma * nv * nonterm - val_b[t]
                adv[t] = last = delta + args.gamma * args.lam * nonterm * last
            ret = adv + val_b
        b_obs, b_act = obs_b.reshape(-1, obs_dim), act_b.reshape(-1, act_dim)
        b_logp, b_adv, b_ret = logp_b.reshape(-1), adv.reshape(-1), ret.reshape(-1)
        B = N * T
        mb = B // args.minibatches
        for _ in range(args.epochs):
            perm = torch.randperm(B, device=dev)
            for s in range(0, B, mb):
                idx = perm[s:s + mb]
                d = agent.dist(b_obs[idx])
                ratio = (d.log_prob(b_act[idx]).sum(1) - b_logp[idx]).exp()
                a_mb = b_adv[idx]
                a_mb = (a_mb - a_mb.mean()) / (a_mb.std() + 1e-8)
                pg = torch.max(-a_mb * ratio, -a_mb * ratio.clamp(1 - args.clip, 1 + args.clip)).mean()
                v_loss = 0.5 * ((agent.critic(b_obs[idx]).flatten() - b_ret[idx]) ** 2).mean()
                loss = pg + args.vf_coef * v_loss
                opt.zero_grad(set_to_none=True)
                loss.backward()
                nn.utils.clip_grad_norm_(agent.parameters(), args.max_grad_norm)
                opt.step()
        # one host read per update: mean reward per step of this rollout, finished-episode statistics, distance to the set-point
        stats = torch.stack([rew_b.mean(), fin_ret_sum / fin_n.clamp(min=1), fin_len_sum / fin_n.clamp(min=1), fin_n,
                             next_obs[:, 0:3].norm(dim=1).mean()]).tolist()
        fin_ret_sum.zero_(), fin_len_sum.zero_(), fin_n.zero_()
        sim_time += ev0.elapsed_time(ev1) * 1e-3
        wall = time.time() - t0
        log.append({"update": update, "wall_s": round(wall, 3), "env_steps": env_steps, "mean_reward_per_step": stats[0],
                    "mean_return_of_finished_episodes": stats[1], "mean_length_of_finished_episodes": stats[2],
                    "finished_episodes": int(stats[3]), "mean_distance_to_setpoint_m": stats[4]})
        if update % 10 == 0 or wall > args.seconds:
            print(f"update {update:4d}  {wall:6.1f} s  {env_steps / 1e6:7.1f} M env-steps  reward/step {stats[0]:6.2f}  "
                  f"episode return {stats[1]:8.1f}  distance {stats[4]:.3f} m", flush=True)
        if wall > args.seconds:
            break
    wall = time.time() - t0

    def first(key, thr):
        for row in log:
            if row[key] >= thr:
                return row["wall_s"]
        return None

    out = {"what": "PPO (examples/train_ppo_position.py; hyper-parameters of the reference's cleanrl script) on position_setpoint_task, "
                   f"{N} envs, {cfg.controller_name}, one MI355X, everything on the device",
           "reference_claim": "docs/6_rl_training.md:134-140: 'trains in under a minute using a single NVIDIA RTX 3090' (rl_games, same task)",
           "wall_s": wall, "updates": len(log), "env_steps": env_steps, "env_steps_per_s_incl_learning": env_steps / wall,
           "rollout_share_of_wall": sim_time / wall,
           "seconds_to_mean_reward_per_step": {str(th): first("mean_reward_per_step", th) for th in (5, 10, 15, 20, 25)},
           "seconds_to_mean_episode_return": {str(th): first("mean_return_of_finished_episodes", th) for th in (2500, 5000, 7500, 10000)},
           "final": log[-1], "max_reward_per_step_possible": 30.5, "curve": log[:: max(1, len(log) // 200)]}
    print(json.dumps({k: v for k, v in out.items() if k != "curve"}, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
