"""TEST INFRASTRUCTURE -- golden vectors for SURVEY.md section 8 row f2 (LiDAR navigation task),
produced by running the REFERENCE's own code (build container only: needs /root/reference).

    python oracle/gen_golden_lidar_nav.py     # writes tests/golden/{robot_magpie,step_magpie_acceleration,
                                              #   reward_lidar_navigation,lidar_image_obs,obs_lidar_navigation}.npz

  robot_magpie.npz             composite mass / inertia of resources/robots/magpie/model.urdf
  step_magpie_acceleration.npz BaseMultirotor.step with MagpieCfg + magpie_acceleration_control
                               (force_application_level "base_link": the wrench A u acts on the root body)
  reward_lidar_navigation.npz  lidar_navigation_task.compute_reward (:554-719)
  lidar_image_obs.npz          LiDARNavigationTask.process_image_observation (:313-363) and
                               add_noise_to_downsampled_lidar_data (:281-310), called as unbound methods on a
                               stand-in object; the random tensors the noise function draws are re-drawn
                               from the same seed and stored next to the outputs
  obs_lidar_navigation.npz     LiDARNavigationTask.process_obs_for_task (:440-470), same technique
"""
import json
import math
import os
import types

import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402  FIRST: it switches TorchScript off before torch is imported (cr_torch.py)

import numpy as np
import torch

import oracle as orc
import ref_shells

OUT = gg.OUT


def magpie_constants(cfg):
    links = gg.parse_urdf(os.path.join(ref_shells.REFERENCE_ROOT, "resources", "robots", "magpie", "model.urdf"))
    mass, com, J = gg.composite(links)
    props = [n for n in links if n != "base_link"]
    return dict(mass=np.float64(mass), com=com, inertia=J, alloc=np.array(cfg.control_allocator_config.allocation_matrix, np.float64),
                wrench_map=np.array(cfg.control_allocator_config.allocation_matrix, np.float64),  # root mode: w = A u
                base_mass=np.float64(links["base_link"]["mass"]), base_inertia=links["base_link"]["inertia"],
                motor_mass=np.float64(links[props[0]]["mass"]), motor_inertia=links[props[0]]["inertia"],
                motor_pos=np.array([links[p]["xyz"] for p in props]), collision_radius=np.float64(0.35))


def root_mode_constants(folder, cfg, collision_radius):
    """composite rigid body of resources/robots/<folder>/model.urdf + the config's allocation matrix (root-link allocators)"""
    links = gg.parse_urdf(os.path.join(ref_shells.REFERENCE_ROOT, "resources", "robots", folder, "model.urdf"))
    mass, com, J = gg.composite(links)
    props = [n for n in links if n != "base_link"]
    ca = cfg.control_allocator_config
    mm = ca.motor_model_config
    return dict(mass=np.float64(mass), com=com, inertia=J, alloc=np.array(ca.allocation_matrix, np.float64),
                base_mass=np.float64(links["base_link"]["mass"]), base_inertia=links["base_link"]["inertia"],
                motor_mass=np.float64(links[props[0]]["mass"]), motor_pos=np.array([links[p]["xyz"] for p in props]),
                collision_radius=np.float64(collision_radius), force_application_level=np.array(ca.force_application_level),
                motor_model=np.array([mm.motor_thrust_constant_min, mm.motor_thrust_constant_max, mm.motor_time_constant_increasing_min,
                                      mm.motor_time_constant_increasing_max, mm.motor_time_constant_decreasing_min,
                                      mm.motor_time_constant_decreasing_max, mm.max_thrust, mm.min_thrust, mm.max_thrust_rate,
                                      mm.thrust_to_torque_ratio], np.float64),
                min_init_state=np.array(cfg.init_config.min_init_state, np.float64), max_init_state=np.array(cfg.init_config.max_init_state, np.float64),
                disturbance=np.array([float(cfg.disturbance.enable_disturbance), cfg.disturbance.prob_apply_disturbance]
                                     + list(cfg.disturbance.max_force_and_torque_disturbance), np.float64))


def controller_gain_table(ctrl):
    return np.array([ctrl.K_pos_tensor_min, ctrl.K_pos_tensor_max, ctrl.K_vel_tensor_min, ctrl.K_vel_tensor_max, ctrl.K_rot_tensor_min,
                     ctrl.K_rot_tensor_max, ctrl.K_angvel_tensor_min, ctrl.K_angvel_tensor_max], np.float64)


def gen_step_root_mode(cfg, consts, n=64, K=6, seed=11):
    """gen_golden.gen_step for a robot whose allocator applies the combined wrench at the root body."""
    rng = torch.Generator().manual_seed(seed)
    robot, gtd = gg.make_ref_robot(cfg, "magpie_acceleration_control", n, consts, seed)
    ctrl_cfg = robot.controller_config
    robot.controller.randomize_params(torch.arange(n))  # magpie_controller_config.randomize_params = True
    M, A = cfg.control_allocator_config.num_motors, robot.num_actions
    pd = gg.params_dict(cfg, ctrl_cfg, "acceleration", consts)
    pd["root_link_mode"] = 1  # control_allocation.py:53-65: anything but "motor_link"
    P = orc.make_params(pd)
    gtd["robot_state_tensor"][:] = gg.random_state(n, rng)
    thrust, kT, tinc, tdec = gg.motor_arrays(robot, n, M)
    Kp, Kv, KR, Kw = gg.gains(robot, n)
    keys = ("state", "action", "thrust_in", "thrust_out", "euler", "qveh", "vveh", "vbody", "wbody", "wrench_cmd", "force",
            "torque", "disturb", "action_after")
    rec = {k: [] for k in keys}
    dmax = torch.tensor(cfg.disturbance.max_force_and_torque_disturbance)
    assert robot.application_mask.tolist() == [0]
    for k in range(K):
        action = (torch.rand(n, A, generator=rng) - 0.5) * 2 * 2.0  # the task scales accelerations to +-2 m/s^2
        if k == K - 1:
            action = action * 30.0
        rec["state"].append(gtd["robot_state_tensor"].clone())
        rec["action"].append(action.clone())
        rec["thrust_in"].append(thrust.clone())
        sd = 2000 + 13 * k
        torch.manual_seed(sd)
        robot.step(action.clone())
        d = torch.zeros(n, 7)
        torch.manual_seed(sd)  # replay apply_disturbance's draws (base_multirotor.py:213-234)
        d[:, 0] = torch.bernoulli(cfg.disturbance.prob_apply_disturbance * torch.ones(n))
        d[:, 1:4] = torch.rand_like(dmax[0:3].expand(n, -1))
        d[:, 4:7] = torch.rand_like(dmax[3:6].expand(n, -1))
        rec["disturb"].append(d)
        rec["thrust_out"].append(thrust.clone())
        rec["action_after"].append(robot.action_tensor.clone())
        for name, t in (("euler", robot.robot_euler_angles), ("qveh", robot.robot_vehicle_orientation),
                        ("vveh", robot.robot_vehicle_linvel), ("vbody", robot.robot_body_linvel), ("wbody", robot.robot_body_angvel),
                        ("wrench_cmd", robot.controller.wrench_command), ("force", gtd["robot_force_tensor"]),
                        ("torque", gtd["robot_torque_tensor"])):
            rec[name].append(t.clone())
        bw = np.concatenate([gtd["robot_force_tensor"][:, 0, :].numpy(), gtd["robot_torque_tensor"][:, 0, :].numpy()], axis=1)
        st = np.ascontiguousarray(gtd["robot_state_tensor"].numpy().astype(np.float32))
        orc.integrate(P, st, np.ascontiguousarray(bw.astype(np.float32)))
        gtd["robot_state_tensor"][:] = torch.from_numpy(st)
    out = {k: torch.stack(v).numpy() for k, v in rec.items()}
    assert np.abs(out["force"][:, :, 1:]).max() == 0.0  # nothing is applied to the prop bodies
    out.update(kT=kT.numpy(), tau_inc=tinc.numpy(), tau_dec=tdec.numpy(), Kp=Kp.numpy(), Kv=Kv.numpy(), KR=KR.numpy(),
               Kw=Kw.numpy(), disturb_max=dmax.numpy(), application_mask=np.array([0]), params_json=np.array(json.dumps(pd)))
    np.savez(os.path.join(OUT, "step_magpie_acceleration.npz"), **out)
    print("step_magpie_acceleration: ok  wrench[0] =", out["wrench_cmd"][0, 0])


REWARD_KEYS = ["pos_reward_magnitude", "pos_reward_exponent", "very_close_to_goal_reward_magnitude",
               "very_close_to_goal_reward_exponent", "vel_direction_component_reward_magnitude",
               "x_action_diff_penalty_magnitude", "x_action_diff_penalty_exponent", "y_action_diff_penalty_magnitude",
               "y_action_diff_penalty_exponent", "z_action_diff_penalty_magnitude", "z_action_diff_penalty_exponent",
               "yawrate_action_diff_penalty_magnitude", "yawrate_action_diff_penalty_exponent",
               "x_absolute_action_penalty_magnitude", "x_absolute_action_penalty_exponent",
               "y_absolute_action_penalty_magnitude", "y_absolute_action_penalty_exponent",
               "z_absolute_action_penalty_magnitude", "z_absolute_action_penalty_exponent",
               "yawrate_absolute_action_penalty_magnitude", "yawrate_absolute_action_penalty_exponent", "collision_penalty"]


def gen_reward(rng, lt, cfg):
    n = 768
    pe = torch.randn(n, 3, generator=rng) * 3
    pe[: n // 6] *= 0.15  # inside the 1 m "stable at goal" radius
    prev = pe + torch.randn(n, 3, generator=rng) * 0.1
    vveh = torch.randn(n, 3, generator=rng) * 1.5
    vveh[::7] *= 3.0  # beyond the 3 m/s penalty knee
    wbody = torch.randn(n, 3, generator=rng)
    yaw_err = (torch.rand(n, generator=rng) - 0.5) * 2 * math.pi
    crashes = torch.rand(n, generator=rng) < 0.1
    act = (torch.rand(n, 4, generator=rng) - 0.5) * 4
    pact = (torch.rand(n, 4, generator=rng) - 0.5) * 4
    ttc = torch.rand(n, generator=rng) * 3
    ttc[::5] = 10.0
    cpf = 0.35
    pdct = {k: torch.tensor(float(cfg.reward_parameters[k])) for k in REWARD_KEYS}
    r, c = lt.compute_reward(pe, prev, vveh, wbody, yaw_err, crashes.clone(), act, pact, ttc, cpf, pdct)
    np.savez(os.path.join(OUT, "reward_lidar_navigation.npz"), pos_err=pe.numpy(), prev_pos_err=prev.numpy(), vveh=vveh.numpy(),
             wbody=wbody.numpy(), yaw_error=yaw_err.numpy(), crashes=crashes.numpy(), action=act.numpy(), prev_action=pact.numpy(),
             time_to_collision=ttc.numpy(), curriculum_progress=np.float32(cpf),
             rp=np.array([float(cfg.reward_parameters[k]) for k in REWARD_KEYS], np.float32), reward=r.numpy())
    cfg.device = "cpu"
    ain = (torch.rand(64, 4, generator=rng) * 3 - 1.5)
    aout = cfg.action_transformation_function(ain.clone())
    d = dict(np.load(os.path.join(OUT, "reward_lidar_navigation.npz")))
    d.update(action_transform_in=ain.numpy(), action_transform_out=aout.numpy())
    np.savez(os.path.join(OUT, "reward_lidar_navigation.npz"), **d)
    print("reward_lidar_navigation: ok  mean", float(r.mean()))


def gen_image_obs(rng, lt):
    n, H, W = 6, 48, 120
    m = ref_shells.ref("utils.math")
    pos = (torch.rand(n, 3, generator=rng) - 0.5) * 4
    # a world-frame point cloud as the sensor writes it: hits at 0.1 .. 12 m, misses at 1000 m along the ray
    dirs = torch.randn(n, H, W, 3, generator=rng)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    rng_img = torch.rand(n, H, W, generator=rng) * 12 + 0.05
    rng_img[torch.rand(n, H, W, generator=rng) < 0.15] = 1000.0
    pc = (pos[:, None, None, :] + dirs * rng_img[..., None]).unsqueeze(1).contiguous()
    linvel = torch.randn(n, 3, generator=rng) * 2
    linvel[0] = 0.0
    out = {}
    for tag, seed in (("clean", None), ("noisy", 77)):
        fake = types.SimpleNamespace(
            obs_dict={"depth_range_pixels": pc.clone(), "robot_position": pos.clone(), "robot_linvel": linvel.clone()},
            world_dir_vectors=torch.ones(n, H, W, 3), num_envs=n, time_to_collision=torch.zeros(n),
            downsampled_lidar_data=torch.zeros(n, 16 * 20), device="cpu")
        if seed is None:
            fake.add_noise_to_downsampled_lidar_data = lambda x: x
        else:
            fake.add_noise_to_downsampled_lidar_data = types.MethodType(lt.LiDARNavigationTask.add_noise_to_downsampled_lidar_data, fake)
            torch.manual_seed(seed)
        lt.LiDARNavigationTask.process_image_observation(fake)
        out[tag + "_ttc"] = fake.time_to_collision.numpy().copy()
        out[tag + "_ds"] = fake.downsampled_lidar_data.numpy().copy()
        if seed is not None:
            # re-draw what add_noise_to_downsampled_lidar_data drew (:281-310), same seed, same call order
            torch.manual_seed(seed)
            z = torch.zeros(n, 16, 20)
            noise_mask = torch.bernoulli(0.03 * torch.ones_like(z))
            k1 = int((noise_mask == 1).sum())
            noise_val_flat = m.torch_rand_float_tensor(0.2 * torch.ones(k1), 10.0 * torch.ones(k1))
            noise_val = torch.zeros_like(z)
            noise_val[noise_mask == 1] = noise_val_flat
            max_mask = torch.bernoulli(0.02 * torch.ones_like(z))
            low_mask = torch.bernoulli(0.02 * torch.ones_like(z[:, 10:]))
            low_val = m.torch_rand_float_tensor(0.2 * torch.ones_like(low_mask), 1.0 * torch.ones_like(low_mask))
            out.update(noise_mask=noise_mask.numpy(), noise_val=noise_val.numpy(), max_mask=max_mask.numpy(),
                       low_mask=low_mask.numpy(), low_val=low_val.numpy())
    np.savez(os.path.join(OUT, "lidar_image_obs.npz"), pointcloud=pc.numpy(), robot_position=pos.numpy(), robot_linvel=linvel.numpy(), **out)
    print("lidar_image_obs: ok  ttc", out["clean_ttc"], " noisy != clean:", int((out["noisy_ds"] != out["clean_ds"]).sum()))


def gen_obs(rng, lt):
    n = 96
    m = ref_shells.ref("utils.math")
    state = gg.random_state(n, rng, spread=5.0)
    qveh = m.vehicle_frame_quat_from_quat(state[:, 3:7])
    euler = m.get_euler_xyz_tensor(state[:, 3:7])
    target = (torch.rand(n, 3, generator=rng) - 0.5) * 10
    tyaw = (torch.rand(n, generator=rng) - 0.5) * 2 * math.pi
    vbody, wbody = torch.randn(n, 3, generator=rng), torch.randn(n, 3, generator=rng)
    actions = torch.randn(n, 4, generator=rng)
    ds = torch.rand(n, 320, generator=rng) * 5
    fake = types.SimpleNamespace(
        obs_dict={"robot_vehicle_orientation": qveh, "robot_position": state[:, 0:3], "robot_euler_angles": euler,
                  "robot_body_linvel": vbody, "robot_body_angvel": wbody, "robot_actions": actions},
        target_position=target, target_yaw=tyaw, task_obs={"observations": torch.zeros(n, 337)}, downsampled_lidar_data=ds)
    torch.manual_seed(5)
    lt.LiDARNavigationTask.process_obs_for_task(fake)
    torch.manual_seed(5)
    u_vec = torch.rand_like(target)
    u_euler = torch.rand_like(euler)
    np.savez(os.path.join(OUT, "obs_lidar_navigation.npz"), state=state.numpy(), qveh=qveh.numpy(), euler=euler.numpy(),
             target=target.numpy(), target_yaw=tyaw.numpy(), vbody=vbody.numpy(), wbody=wbody.numpy(), actions=actions.numpy(),
             downsampled=ds.numpy(), u_vec=u_vec.numpy(), u_euler=u_euler.numpy(), obs=fake.task_obs["observations"].numpy())
    print("obs_lidar_navigation: ok")


def main():
    rng = torch.Generator().manual_seed(4242)
    ref_shells.install()
    ref_shells.install_task_shells()
    from aerial_gym.config.robot_config.magpie_config import MagpieCfg
    from aerial_gym.config.task_config.lidar_navigation_task_config import task_config as cfg

    consts = magpie_constants(MagpieCfg)
    np.savez(os.path.join(OUT, "robot_magpie.npz"), **consts)
    print("magpie mass", consts["mass"], "J diag", np.diag(consts["inertia"]), "com", consts["com"])
    gen_step_root_mode(MagpieCfg, consts)
    # the robot / controller of the reference's DEFAULT navigation recipe (navigation_task_config.py:9-10) and the root-link quad
    LMF2Cfg = ref_shells.ref("config.robot_config.lmf2_config").LMF2Cfg
    lmf2_ctrl = ref_shells.ref("config.controller_config.lmf2_controller_config").control
    np.savez(os.path.join(OUT, "robot_lmf2.npz"), gains=controller_gain_table(lmf2_ctrl), randomize_params=np.array(lmf2_ctrl.randomize_params),
             **root_mode_constants("lmf2", LMF2Cfg, 0.25))
    RootCfg = ref_shells.ref("config.robot_config.base_quad_root_link_control_config").BaseQuadRootLinkControlCfg
    np.savez(os.path.join(OUT, "robot_base_quad_root_link_control.npz"), **root_mode_constants("quad", RootCfg, 0.2))
    lt = ref_shells.ref("task.lidar_navigation_task.lidar_navigation_task")
    gen_reward(rng, lt, cfg)
    gen_image_obs(rng, lt)
    gen_obs(rng, lt)


if __name__ == "__main__":
    main()
