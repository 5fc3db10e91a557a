"""TEST / MEASUREMENT INFRASTRUCTURE -- the reference's own CPU path timed on the build container.

SURVEY 8d baseline: "reference torch code imported on CPU + restated integrator".  Runs the REFERENCE's
BaseMultirotor.step (update_states, clip, Lee position controller, allocation, motor model, drag, disturbance:
robots/base_multirotor.py:296-307 + control/**), the oracle's rigid-body integrator in place of Isaac Gym's CPU
PhysX (not installable), and the reference's compute_reward (position_setpoint_task.py:245-282) for BASELINE
configs[0] (64 envs) and configs[1] (8192 envs), torch.set_num_threads(nproc).  /root/reference exists only here, so the
result is written to profiles/r04_cpu_baseline_reference.json and echoed by bench.py (`cpu_baseline_reference`); where the
reference tree IS present on the bench box (AERIAL_GYM_REFERENCE_ROOT), bench.py runs this script there and reports it as `cpu_baseline`.

    python oracle/time_reference_cpu.py
"""
import json
import os
import sys
import time

os.environ["AGX_GOLDEN_JIT"] = "1"  # the reference as it ships: TorchScript ON (the golden generators switch it off, cr_torch.py)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
import oracle as orc  # noqa: E402
import ref_shells  # noqa: E402


def run(n, seconds, consts):
    from aerial_gym.config.robot_config.base_quad_config import BaseQuadCfg

    ref_shells.install_task_shells()
    pt = ref_shells.ref("task.position_setpoint_task.position_setpoint_task")
    m = ref_shells.ref("utils.math")
    robot, gtd = G.make_ref_robot(BaseQuadCfg, "lee_position_control", n, consts, 1)
    pd = G.params_dict(BaseQuadCfg, robot.controller_config, "position", consts)
    P = orc.make_params(pd)
    mask = torch.tensor(BaseQuadCfg.control_allocator_config.application_mask)
    W = consts["wrench_map"].astype(np.float32)
    robot.reset_idx(torch.arange(n))
    gen = torch.Generator().manual_seed(1234)
    actions = [torch.rand(n, 4, generator=gen) * 2 - 1 for _ in range(8)]
    target = torch.zeros(n, 3)
    crashes = torch.zeros(n, dtype=torch.bool)

    def step(i):
        a = actions[i % 8]
        robot.step(a.clone())
        u = gtd["robot_force_tensor"][:, mask, 2].numpy().astype(np.float32)
        bw = (u @ W.T).astype(np.float32)
        bw[:, 0:3] += gtd["robot_force_tensor"][:, 0, :].numpy()
        bw[:, 3:6] += gtd["robot_torque_tensor"][:, 0, :].numpy()
        st = np.ascontiguousarray(gtd["robot_state_tensor"].numpy())
        orc.integrate(P, st, np.ascontiguousarray(bw))
        gtd["robot_state_tensor"][:] = torch.from_numpy(st)
        pe = m.quat_apply_inverse(robot.robot_vehicle_orientation, target - gtd["robot_position"])
        crashes[:] = False
        pt.compute_reward(pe, gtd["robot_linvel"], gtd["robot_orientation"], robot.robot_body_angvel, crashes, 1.0, a, a,
                          {"x": torch.zeros(1)})

    for i in range(20):
        step(i)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for i in range(10):
            step(i)
        steps += 10
    dt = time.perf_counter() - t0
    return {"num_envs": n, "steps": steps, "seconds": dt, "value": n * steps / dt, "ms_per_step": 1e3 * dt / steps}


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "profiles", "r04_cpu_baseline_reference.json"))
    ap.add_argument("--host", default="build container (the reference tree is not on the GPU box)")
    ap.add_argument("--seconds", type=float, default=15.0, help="budget of the 8192-env leg (the 64-env leg gets half)")
    args = ap.parse_args()
    ref_shells.install()
    from aerial_gym.config.robot_config.base_quad_config import BaseQuadCfg

    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    consts = G.robot_constants("quad", BaseQuadCfg)
    out = {
        "kind": "reference",
        "unit": "env-steps/s",
        "cores": threads,
        "host": args.host,
        "cpu_model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
        "what": "reference BaseMultirotor.step + Lee position controller + allocation + motor model (torch %s, %d threads) + "
                "restated rigid-body integrator (Isaac Gym's CPU PhysX is not installable) + reference compute_reward; no reset, "
                "no observation packing" % (torch.__version__, threads),
        "configs": {"configs[0] 64 envs": run(64, args.seconds / 2, consts), "configs[1] 8192 envs": run(8192, args.seconds, consts)},
    }
    out["value"] = out["configs"]["configs[1] 8192 envs"]["value"]
    out["sample"] = "%d env steps of 8192 envs (%.1f s)" % (out["configs"]["configs[1] 8192 envs"]["steps"],
                                                           out["configs"]["configs[1] 8192 envs"]["seconds"])
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
