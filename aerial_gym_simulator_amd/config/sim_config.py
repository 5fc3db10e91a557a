"""aerial_gym/config/sim_config/base_sim_config.py:19-37 (values consumed by the integrator)."""


class BaseSimConfig:
    class viewer:
        headless = True  # there is no viewer in this package; kept for API compatibility
        ref_env = 0

    class sim:
        dt = 0.01
        substeps = 1
        gravity = [0.0, 0.0, -9.81]
        up_axis = 1
        use_gpu_pipeline = True


class BaseSimHeadlessConfig(BaseSimConfig):  # config/sim_config/base_sim_headless_config.py
    class viewer(BaseSimConfig.viewer):
        headless = True


class BaseSimNoGravityConfig(BaseSimConfig):  # config/sim_config/base_sim_no_gravity_config.py (examples/imu_data_collection.py:8)
    class sim(BaseSimConfig.sim):
        gravity = [0.0, 0.0, 0.0]


class SimCfg2Ms(BaseSimConfig):  # config/sim_config/sim_config_2ms.py:21-22 (the PhysX solver block has no counterpart here)
    class sim(BaseSimConfig.sim):
        dt = 0.002


class SimCfg4Ms(BaseSimConfig):  # config/sim_config/sim_config_4ms.py:21-22
    class sim(BaseSimConfig.sim):
        dt = 0.004
