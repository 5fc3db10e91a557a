"""A robot CLASS as a plug-in (the reference's contract: robots/base_robot.py:10-63, robot_manager.py:486-489).

    python examples/custom_robot.py        (on an MI355X)

`LiftAssistedQuad.step(action)` is called by the EnvManager once per physics sub-step.  `super().step(action)` is the reference's
BaseMultirotor.step as ONE launch (update_states, clip, Lee controller, allocation + motor model, the per-body force / torque tensors,
drag, disturbance); whatever the class then leaves in `robot_force_tensors` / `robot_torque_tensors` -- each body's wrench in that
body's own frame, as Isaac Gym applies them in the reference -- is reduced to the net wrench on the rigid composite and integrated.
A robot whose class leaves `step` alone keeps the fused one-launch path; see INTEGRATION.md section 3 for the cost of the plug-in path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aerial_gym_simulator_amd  # noqa: E402,F401
from aerial_gym_simulator_amd.registry.robot_registry import robot_registry  # noqa: E402
from aerial_gym_simulator_amd.robots.base_multirotor import BaseMultirotor  # noqa: E402
from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder  # noqa: E402


class LiftAssistedQuad(BaseMultirotor):
    """a quadrotor under a tethered balloon: a constant world-frame lift at the base link on top of what the motors do"""

    LIFT_N = 0.6

    def step(self, action_tensor):
        super().step(action_tensor)
        # world +z expressed in the base link's frame: R^T e_z (robot_orientation is xyzw)
        q = self.robot_orientation
        x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        ez_body = torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
        self.robot_force_tensors[:, 0, :] += self.LIFT_N * ez_body


def main(num_envs=1024, steps=200, device="cuda:0"):
    robot_registry.register("lift_assisted_quadrotor", LiftAssistedQuad, robot_registry.get_robot_config("base_quadrotor"))
    env = SimBuilder().build_env(sim_name="base_sim", env_name="empty_env", robot_name="lift_assisted_quadrotor",
                                 controller_name="lee_position_control", device=device, num_envs=num_envs, headless=True, use_warp=False)
    assert env.robot_manager.robot.external_robot
    env.reset()
    g = env.get_obs()
    target = torch.zeros(num_envs, 4, device=device)  # hold the origin, yaw 0
    for _ in range(steps):
        env.step(actions=target)
    torch.cuda.synchronize()
    p = g["robot_position"]
    print(f"{num_envs} lift-assisted quadrotors after {steps} steps of position hold at the origin: mean z {float(p[:, 2].mean()):+.3f} m "
          f"(a proportional position law settles above its set-point under {LiftAssistedQuad.LIFT_N} N of extra lift), "
          f"mean |xy| {float(p[:, :2].norm(dim=1).mean()):.3f} m")


if __name__ == "__main__":
    main()
