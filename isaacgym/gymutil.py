"""isaacgym.gymutil names used by the reference's argument parsing (utils/helpers.py:91-160)."""


def parse_device_str(device_str):
    """'cuda:1' -> ('cuda', 1); 'cpu' -> ('cpu', 0)   (Isaac Gym's own helper, same contract)."""
    device, _, idx = str(device_str).lower().partition(":")
    if device not in ("cpu", "cuda", "gpu"):
        raise ValueError(f"Invalid device string '{device_str}': cpu, cuda or cuda:N")
    if device == "gpu":
        device = "cuda"
    return device, int(idx) if idx else 0


def parse_sim_config(sim_cfg, sim_params):
    """Copies a `sim` config dict onto a SimParams-like object (attribute per key, nested dicts onto nested objects)."""
    for key, val in sim_cfg.items():
        if isinstance(val, dict) and hasattr(sim_params, key):
            parse_sim_config(val, getattr(sim_params, key))
        else:
            setattr(sim_params, key, val)


def parse_arguments(*args, **kwargs):
    from aerial_gym_simulator_amd.utils.helpers import parse_arguments as _pa

    return _pa(*args, **kwargs)
