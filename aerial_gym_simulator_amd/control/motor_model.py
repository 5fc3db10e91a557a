"""Host mirror of aerial_gym/control/motor_model.py: owns the per-env, per-motor tensors
(thrust state and the randomised first-order-model parameters).  The update itself
(clamp, time-constant selection, RK4/Euler in thrust or RPM domain) runs inside
agx_dynamics_substeps (csrc/agx_dynamics.hip: motor_update)."""
from ..tensors import aos_view, soa


class MotorModel:
    def __init__(self, num_envs, motors_per_robot, dt, config, device="cuda:0", random_source=None):
        self.num_envs, self.num_motors_per_robot, self.dt = num_envs, motors_per_robot, dt
        self.cfg, self.device = config, device
        scheme = getattr(config, "integration_scheme", "rk4")
        self.integration_scheme = scheme if scheme in ("euler", "rk4") else "rk4"
        M, N = motors_per_robot, num_envs
        self.thrust_soa = soa(M, N, device)
        self.tau_inc_soa = soa(M, N, device)
        self.tau_dec_soa = soa(M, N, device)
        self.kT_soa = soa(M, N, device, fill=1.0)
        # reference-shaped [N, M] aliases
        self.current_motor_thrust = aos_view(self.thrust_soa)
        self.motor_time_constants_increasing = aos_view(self.tau_inc_soa)
        self.motor_time_constants_decreasing = aos_view(self.tau_dec_soa)
        self.motor_thrust_constant = aos_view(self.kT_soa)
        self.ranges = dict(
            thrust=(float(config.min_thrust), float(config.max_thrust)),
            tau_inc=(float(config.motor_time_constant_increasing_min), float(config.motor_time_constant_increasing_max)),
            tau_dec=(float(config.motor_time_constant_decreasing_min), float(config.motor_time_constant_decreasing_max)),
            kT=(float(config.motor_thrust_constant_min), float(config.motor_thrust_constant_max)),
        )
        if random_source is not None:
            self.init_tensors(random_source)

    def init_tensors(self, rs):
        """Initial draws in the reference's order (motor_model.py:44-82)."""
        N, M = self.num_envs, self.num_motors_per_robot
        for name, view in (("thrust", self.current_motor_thrust), ("tau_inc", self.motor_time_constants_increasing),
                           ("tau_dec", self.motor_time_constants_decreasing)):
            lo, hi = self.ranges[name]
            view[:] = (hi - lo) * rs.rand(N, M, tag="motor_init_" + name) + lo
        if self.cfg.use_rps:
            lo, hi = self.ranges["kT"]
            self.motor_thrust_constant[:] = (hi - lo) * rs.rand(N, M, tag="motor_init_kT") + lo
