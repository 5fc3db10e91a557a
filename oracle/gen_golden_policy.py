"""TEST INFRASTRUCTURE.  Fixture: the actor of a policy the REFERENCE trained in its own simulator (Isaac Gym / PhysX),
`aerial_gym/examples/rl_games_example/networks/attitude_policy.pth` -- the network the reference's closed-loop example
flies (`examples/rl_games_example/rl_env_closed_loop_example.py:41-50`: position_setpoint_task, base_quadrotor,
lee_attitude_control, 13-D observation -> 4-D attitude command).  tests/test_gpu_policy_transfer.py flies it on the HIP
path: a behavioural cross-check of the rigid-body integration + controller stack against an artefact that only ever saw
PhysX (the integrator row of SURVEY.md section 8 has no numeric fixture: PhysX is not in the reference tree).

Stored: the actor MLP 13 -> 256 -> 128 -> 64 -> 4 (ELU; rl_games_inference.py:7-46), its log-std, and the mean episode
return rl_games logged for it (`last_mean_rewards`).  Weights only -- no reference code.

    python oracle/gen_golden_policy.py [/root/reference]
"""
import os
import sys

import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402  FIRST: it switches TorchScript off before torch is imported (cr_torch.py)

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

OUT = gg.OUT


# checkpoint -> fixture.  The two lmf2 networks are the ones the reference deploys on the real vehicle
# (examples/rl_games_example/rl_games_ros_node.py:17-27), trained in position_setpoint_task_acceleration_sim2real /
# position_setpoint_task_sim2real (lmf2 + lmf2_acceleration_control / lmf2_velocity_control, 17-D observation = position error,
# quaternion, body-frame velocities, previous action): tests/test_gpu_policy_transfer.py flies them too.
POLICIES = {
    "attitude_policy.pth": "policy_attitude_actor.npz",
    "acc_command_2_multiplier_disturbance.pth": "policy_lmf2_acceleration_actor.npz",
    "vel_control_lmf2_direct.pth": "policy_lmf2_velocity_actor.npz",
}


def main(ref="/root/reference"):
    for src, name in POLICIES.items():
        path = os.path.join(ref, "aerial_gym/examples/rl_games_example/networks", src)
        ck = torch.load(path, map_location="cpu", weights_only=False)
        m = ck["model"]
        out = {}
        for i, layer in enumerate(("actor_mlp.0", "actor_mlp.2", "actor_mlp.4", "mu")):
            out[f"w{i}"] = m[f"a2c_network.{layer}.weight"].numpy().astype(np.float32)
            out[f"b{i}"] = m[f"a2c_network.{layer}.bias"].numpy().astype(np.float32)
        out["logstd"] = m["a2c_network.sigma"].numpy().astype(np.float32)
        out["last_mean_rewards"] = np.float64(float(ck["last_mean_rewards"]))
        out["epoch"], out["frame"] = np.int64(ck["epoch"]), np.int64(ck["frame"])
        dst = os.path.join(OUT, name)
        np.savez_compressed(dst, **out)
        print(dst, {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main(*sys.argv[1:])
