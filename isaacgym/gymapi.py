"""isaacgym.gymapi names the reference's helpers read (utils/helpers.py:56-88, 150-156).  No simulator behind them."""
SIM_PHYSX = 0
SIM_FLEX = 1
UP_AXIS_Y, UP_AXIS_Z = 0, 1
DOF_MODE_NONE, DOF_MODE_POS, DOF_MODE_VEL, DOF_MODE_EFFORT = 0, 1, 2, 3
LOCAL_SPACE, ENV_SPACE, GLOBAL_SPACE = 0, 1, 2


class _Bag:
    def __repr__(self):
        return f"{type(self).__name__}({self.__dict__})"


class PhysXParams(_Bag):
    def __init__(self):
        self.use_gpu, self.num_subscenes, self.num_threads = True, 0, 0
        self.solver_type, self.num_position_iterations, self.num_velocity_iterations = 1, 4, 1


class SimParams(_Bag):
    def __init__(self):
        self.dt, self.substeps, self.use_gpu_pipeline = 0.01, 1, True
        self.up_axis, self.gravity = UP_AXIS_Z, (0.0, 0.0, -9.81)
        self.physx = PhysXParams()


class Vec3(_Bag):
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = x, y, z


class AssetOptions(_Bag):
    pass


def acquire_gym(*a, **k):
    raise RuntimeError("isaacgym is an inert stand-in here: aerial_gym_simulator_amd simulates in its own HIP kernels "
                       "(use aerial_gym.sim.sim_builder.SimBuilder / task_registry.make_task)")
