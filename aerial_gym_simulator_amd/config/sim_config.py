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
