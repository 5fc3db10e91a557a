"""Test helper: drives the C ABI of libaerialgym_hip.so directly from numpy data.

AoS reference-layout arrays ([N, C]) are uploaded as the SoA buffers the library expects;
results come back as AoS numpy arrays so they can be compared 1:1 with the oracle."""
import ctypes as C

import numpy as np
import torch

from aerial_gym_simulator_amd import _lib
from aerial_gym_simulator_amd._lib import AgxEnvBuffers, AgxResetArgs
from aerial_gym_simulator_amd.robots.robot_model import pack_robot_params

CTRL_KEY = {"no_control": "none"}


def product_params(pd):
    d = dict(pd)
    d["controller"] = CTRL_KEY.get(pd["controller"], pd["controller"])
    return pack_robot_params(d)


def to_soa(a, dev):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32).T)).to(dev).contiguous()


def to_aos(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().T)


class DynHarness:
    def __init__(self, pd, n, dev="cuda:0"):
        self.lib = _lib.load()
        self.P = product_params(pd)
        self.pd, self.n, self.dev = pd, n, dev
        M, A = pd["num_motors"], pd["num_actions"]
        z = lambda c, dt=torch.float32: torch.zeros(c, n, dtype=dt, device=dev)  # noqa: E731
        self.t = dict(state=z(13), derived=z(16), actions=z(A), prev_actions=z(A), thrust=z(M), kT=z(M) + 1, tau_inc=z(M),
                      tau_dec=z(M), gains=z(12), wrench=z(6), bmin=z(3) - 1, bmax=z(3) + 1)
        self.crashes = torch.zeros(n, dtype=torch.bool, device=dev)
        self.trunc = torch.zeros(n, dtype=torch.bool, device=dev)
        self.sim_steps = torch.zeros(n, dtype=torch.int32, device=dev)
        self.reset_mask = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.reset_flag = torch.zeros(2, dtype=torch.int32, device=dev)
        self.episode_count = torch.zeros(n, dtype=torch.int32, device=dev)
        self.disturb = None
        self.boxes = None
        self.B = AgxEnvBuffers()
        self.rebind()

    def rebind(self):
        B, t, p = self.B, self.t, _lib.dptr
        B.state, B.derived, B.actions, B.prev_actions = p(t["state"]), p(t["derived"]), p(t["actions"]), p(t["prev_actions"])
        B.motor_thrust, B.motor_kT, B.motor_tau_inc, B.motor_tau_dec = p(t["thrust"]), p(t["kT"]), p(t["tau_inc"]), p(t["tau_dec"])
        B.gains, B.wrench_cmd = p(t["gains"]), p(t["wrench"])
        B.crashes, B.truncations, B.sim_steps = p(self.crashes), p(self.trunc), p(self.sim_steps)
        B.reset_mask, B.reset_flag = p(self.reset_mask), p(self.reset_flag)
        B.flag_parity = 0
        B.episode_count = p(self.episode_count)
        B.bounds_min, B.bounds_max = p(t["bmin"]), p(t["bmax"])
        B.disturb = p(self.disturb) if self.disturb is not None else None
        B.boxes = p(self.boxes) if self.boxes is not None else None
        B.num_boxes = 0 if self.boxes is None else self.boxes.shape[0] // 11

    def stream(self):
        return _lib.current_stream(self.dev)

    def set(self, **arrays):
        for k, a in arrays.items():
            self.t[k].copy_(to_soa(a, self.dev))

    def set_gains(self, Kp, Kv, KR, Kw):
        self.t["gains"].copy_(to_soa(np.concatenate([Kp, Kv, KR, Kw], axis=1), self.dev))

    def get(self, name):
        return to_aos(self.t[name])

    def set_disturb(self, d_k, dmax):
        """d_k: [k, N, 7] -> SoA [k][7][N]"""
        self.disturb = torch.from_numpy(np.ascontiguousarray(np.transpose(np.asarray(d_k, np.float32), (0, 2, 1)))).to(self.dev)
        for i in range(6):
            self.B.disturb_max[i] = float(dmax[i])
        self.rebind()

    def set_boxes(self, boxes):
        """boxes: [N, K, 10] -> SoA [K*11][N] (11th channel: bounding radius of the half extents)"""
        b = np.asarray(boxes, np.float32)
        n, k, _ = b.shape
        rad = (np.sqrt((b[..., 7:10].astype(np.float32) ** 2).sum(-1, dtype=np.float32)) * np.float32(1.000001)).astype(np.float32)
        b = np.concatenate([b, rad[..., None]], axis=-1)
        self.boxes = torch.from_numpy(np.ascontiguousarray(b.reshape(n, k * 11).T)).to(self.dev)
        self.rebind()

    def substeps(self, action, k):
        a = torch.from_numpy(np.ascontiguousarray(action, np.float32)).to(self.dev)
        _lib.check(self.lib.agx_dynamics_substeps(self.P, self.B, self.n, _lib.dptr(a), k, self.stream()), "dyn")
        torch.cuda.synchronize()

    def robot_step(self, action, num_bodies, body_of_motor, substep=0):
        """agx_robot_step: BaseMultirotor.step as one launch -> (force [N, B, 3], torque [N, B, 3]) as the reference lays them out"""
        a = torch.from_numpy(np.ascontiguousarray(action, np.float32)).to(self.dev)
        F = torch.zeros(self.n, num_bodies, 3, device=self.dev)
        T = torch.zeros(self.n, num_bodies, 3, device=self.dev)
        args = _lib.AgxRobotStepArgs()
        args.force, args.torque, args.num_bodies, args.substep = _lib.dptr(F), _lib.dptr(T), int(num_bodies), int(substep)
        for j, b in enumerate(body_of_motor):
            args.body_of_motor[j] = int(b)
        _lib.check(self.lib.agx_robot_step(self.P, self.B, self.n, _lib.dptr(a), C.byref(args), self.stream()), "agx_robot_step")
        torch.cuda.synchronize()
        return F.cpu().numpy(), T.cpu().numpy()

    def update_states(self):
        _lib.check(self.lib.agx_update_states(self.B, self.n, self.stream()))
        torch.cuda.synchronize()

    def controller_wrench(self, action):
        a = torch.from_numpy(np.ascontiguousarray(action, np.float32)).to(self.dev)
        _lib.check(self.lib.agx_controller_wrench(self.P, self.B, self.n, _lib.dptr(a), self.stream()))
        torch.cuda.synchronize()
        return self.get("wrench")

    def reward_position(self, target, episode_len, reset_on_collision=1):
        tg = to_soa(target, self.dev)
        rew = torch.zeros(self.n, device=self.dev)
        _lib.check(self.lib.agx_reward_position(self.B, self.n, _lib.dptr(tg), episode_len, reset_on_collision,
                                                _lib.dptr(rew), self.stream()))
        torch.cuda.synchronize()
        return rew.cpu().numpy()

    def obs_position(self, target):
        tg = to_soa(target, self.dev)
        obs = torch.zeros(self.n, 13, device=self.dev)
        _lib.check(self.lib.agx_obs_position(self.B, self.n, _lib.dptr(tg), _lib.dptr(obs), self.stream()))
        torch.cuda.synchronize()
        return obs.cpu().numpy()

    def reward_navigation(self, target, rp, cpf, pos_err, prev_pos_err, episode_len, roc=1):
        tg, pe, ppe = to_soa(target, self.dev), to_soa(pos_err, self.dev), to_soa(prev_pos_err, self.dev)
        rew = torch.zeros(self.n, device=self.dev)
        rp_c = (C.c_float * 18)(*[float(x) for x in rp])
        _lib.check(self.lib.agx_reward_navigation(self.B, self.n, _lib.dptr(tg), rp_c, float(cpf), _lib.dptr(pe), _lib.dptr(ppe),
                                                  episode_len, roc, _lib.dptr(rew), self.stream()))
        torch.cuda.synchronize()
        return rew.cpu().numpy(), to_aos(pe), to_aos(ppe)

    def reset_masked(self, mask, u, ranges, min_state, max_state, bounds_cfg, gains_minmax=None):
        dev, n, M = self.dev, self.n, self.pd["num_motors"]
        self.reset_mask.copy_(torch.from_numpy(np.ascontiguousarray(mask, np.uint8)))
        self.reset_flag.zero_()
        self.reset_flag[0] = 1 if np.any(mask) else 0
        R = AgxResetArgs()
        R.randomize_gains = int(u.get("u_gains") is not None)
        R.seed = int(u.get("seed", 0))
        keep = {}
        if "randomize_gains" in u:
            R.randomize_gains = int(u["randomize_gains"])
        for name in ("u_bounds_lo", "u_bounds_hi", "u_state", "u_gains", "u_tau_inc", "u_tau_dec", "u_thrust", "u_kT"):
            a = u.get(name)
            if a is None:
                setattr(R, name, None)
                continue
            keep[name] = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
            setattr(R, name, _lib.dptr(keep[name]))
        for i in range(3):
            R.lower_bound_min[i], R.lower_bound_max[i] = bounds_cfg[0][i], bounds_cfg[1][i]
            R.upper_bound_min[i], R.upper_bound_max[i] = bounds_cfg[2][i], bounds_cfg[3][i]
        for i in range(13):
            R.min_state[i], R.max_state[i] = float(min_state[i]), float(max_state[i])
        if gains_minmax is not None:
            for i in range(12):
                R.gains_min[i], R.gains_max[i] = float(gains_minmax[0][i]), float(gains_minmax[1][i])
        R.tau_inc_min, R.tau_inc_max = ranges["tau_inc"]
        R.tau_dec_min, R.tau_dec_max = ranges["tau_dec"]
        R.kT_min, R.kT_max = ranges["kT"]
        _lib.check(self.lib.agx_reset_masked(self.P, self.B, self.n, R, self.stream()))
        torch.cuda.synchronize()
