"""TEST INFRASTRUCTURE: the env/task step loop assembled from the CPU oracle's C functions,
following EnvManager.step (env_manager.py:399-432) and PositionSetpointTask.step
(position_setpoint_task.py:152-182) ordering -- see SURVEY.md appendix A."""
import numpy as np

import oracle as orc


class OraclePositionEnv:
    def __init__(self, params, n, episode_len, gains, min_init_state, max_init_state, motor_ranges,
                 bounds=(-1.0, 1.0), substeps=1):
        self.P = orc.make_params(params)
        self.pd = params
        self.n, self.M = n, params["num_motors"]
        self.episode_len, self.substeps = episode_len, substeps
        self.Kp, self.Kv, self.KR, self.Kw = [np.ascontiguousarray(g, np.float32) for g in gains]
        self.min_s = np.asarray(min_init_state, np.float32)
        self.max_s = np.asarray(max_init_state, np.float32)
        self.rng = motor_ranges  # dict name -> (lo, hi)
        self.state = np.zeros((n, 13), np.float32)
        self.state[:, 6] = 1.0
        self.thrust = np.zeros((n, self.M), np.float32)
        self.kT = np.ones((n, self.M), np.float32)
        self.tau_inc = np.zeros((n, self.M), np.float32)
        self.tau_dec = np.zeros((n, self.M), np.float32)
        self.bmin = np.full((n, 3), bounds[0], np.float32)
        self.bmax = np.full((n, 3), bounds[1], np.float32)
        self.target = np.zeros((n, 3), np.float32)
        self.sim_steps = np.zeros(n, np.int32)
        self.euler = np.zeros((n, 3), np.float32)
        self.qveh = np.zeros((n, 4), np.float32)
        self.qveh[:, 3] = 1
        self.vveh = np.zeros((n, 3), np.float32)
        self.vbody = np.zeros((n, 3), np.float32)
        self.wbody = np.zeros((n, 3), np.float32)

    def _lerp(self, name, u):
        lo, hi = self.rng[name]
        return ((np.float32(hi) - np.float32(lo)) * u.astype(np.float32) + np.float32(lo)).astype(np.float32)

    def reset_masked(self, mask, u_state, u_tau_inc, u_tau_dec, u_thrust, u_kT):
        mask = np.ascontiguousarray(mask, np.uint8)
        if not mask.any():
            return
        orc.reset_robot_state(mask, u_state, self.min_s, self.max_s, self.bmin, self.bmax, self.state)
        m = mask.astype(bool)
        self.tau_inc[m] = self._lerp("tau_inc", u_tau_inc)[m]
        self.tau_dec[m] = self._lerp("tau_dec", u_tau_dec)[m]
        self.thrust[m] = self._lerp("thrust", u_thrust)[m]
        if self.pd["use_rps"]:
            self.kT[m] = self._lerp("kT", u_kT)[m]
        self.sim_steps[m] = 0
        # BaseMultirotor.reset_idx ends with update_states() for ALL envs
        self.euler, self.qveh, self.vveh, self.vbody, self.wbody = orc.update_states(self.state)

    def step(self, action, reset_draws=None):
        crashes = np.zeros(self.n, np.uint8)
        for _ in range(self.substeps):
            o = orc.substep(self.P, self.state, action, self.thrust, self.kT, self.tau_inc, self.tau_dec,
                            self.Kp, self.Kv, self.KR, self.Kw)
            self.euler, self.qveh, self.vveh, self.vbody, self.wbody = o.euler, o.qveh, o.vveh, o.vbody, o.wbody
        self.sim_steps += 1
        state_after = self.state.copy()
        reward = orc.reward_position(self.state, self.qveh, self.wbody, self.target, crashes)
        trunc = (self.sim_steps > self.episode_len).astype(np.uint8)
        reset_mask = ((crashes > 0) | (trunc > 0)).astype(np.uint8)
        if reset_mask.any():
            self.reset_masked(reset_mask, *reset_draws)
        obs = orc.obs_position(self.state, self.vbody, self.wbody, self.target)
        return obs, reward, crashes, trunc, reset_mask, state_after
