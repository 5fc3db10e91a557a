"""CLI / config helpers with the reference's names (aerial_gym/utils/helpers.py:56-197),
minus everything that needs isaacgym.gymutil."""
import argparse


def class_to_dict(obj):
    if not hasattr(obj, "__dict__"):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        out[key] = [class_to_dict(v) for v in val] if isinstance(val, list) else class_to_dict(val)
    return out


def update_class_from_dict(obj, dct):
    for key, val in dct.items():
        attr = getattr(obj, key, None)
        if isinstance(attr, type):
            update_class_from_dict(attr, val)
        else:
            setattr(obj, key, val)


def get_args(additional_parameters=None):
    specs = [
        {"name": "--task", "type": str, "default": "position_setpoint_task"},
        {"name": "--experiment_name", "type": str},
        {"name": "--checkpoint", "type": str},
        {"name": "--headless", "action": "store_true", "default": True},
        {"name": "--num_envs", "type": int, "default": 8192},
        {"name": "--seed", "type": int, "default": 1},
        {"name": "--use_warp", "action": "store_true", "default": True},
        {"name": "--sim_device", "type": str, "default": "cuda:0"},
        {"name": "--pipeline", "type": str, "default": "gpu"},
        {"name": "--horovod", "action": "store_true", "default": False},
    ] + list(additional_parameters or [])
    parser = argparse.ArgumentParser(description="aerial_gym_simulator_amd")
    for spec in specs:
        spec = dict(spec)
        name = spec.pop("name")
        parser.add_argument(name, **spec)
    args, _ = parser.parse_known_args()
    args.sim_device_type, _, idx = args.sim_device.partition(":")
    args.sim_device_id = int(idx or 0)
    return args
