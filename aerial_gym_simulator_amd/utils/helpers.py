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


def _strtobool(x):
    v = str(x).strip().lower()
    if v in ("y", "yes", "t", "true", "on", "1"):
        return True
    if v in ("n", "no", "f", "false", "off", "0"):
        return False
    raise ValueError(f"invalid truth value {x!r}")


def parse_device_str(device_str):
    device, _, idx = str(device_str).lower().partition(":")
    if device == "gpu":
        device = "cuda"
    if device not in ("cpu", "cuda"):
        raise ValueError(f"Invalid device string '{device_str}'")
    return device, int(idx) if idx else 0


def parse_arguments(description="Isaac Gym Example", headless=False, no_graphics=False, custom_parameters=[]):  # noqa: B006
    """Same signature, flags and derived fields as the reference's parse_arguments (utils/helpers.py:91-160, itself a
    copy of isaacgym.gymutil.parse_arguments): --sim_device, --pipeline, --graphics_device_id, --flex / --physx,
    --num_threads, --subscenes, --slices, then `custom_parameters` (dicts with name + type/action [+ default, help]);
    unknown arguments are tolerated; adds sim_device_type, compute_device_id, use_gpu_pipeline, physics_engine, use_gpu."""
    parser = argparse.ArgumentParser(description=description)
    if headless:
        parser.add_argument("--headless", action="store_true", help="Run headless without creating a viewer window")
    if no_graphics:
        parser.add_argument("--nographics", action="store_true", help="Disable graphics context creation")
    parser.add_argument("--sim_device", type=str, default="cuda:0", help="Physics Device in PyTorch-like syntax")
    parser.add_argument("--pipeline", type=str, default="gpu", help="Tensor API pipeline (cpu/gpu)")
    parser.add_argument("--graphics_device_id", type=int, default=0, help="Graphics Device ID")
    physics_group = parser.add_mutually_exclusive_group()
    physics_group.add_argument("--flex", action="store_true", help="Use FleX for physics")
    physics_group.add_argument("--physx", action="store_true", help="Use PhysX for physics")
    parser.add_argument("--num_threads", type=int, default=0, help="Number of cores used by PhysX")
    parser.add_argument("--subscenes", type=int, default=0, help="Number of PhysX subscenes to simulate in parallel")
    parser.add_argument("--slices", type=int, help="Number of client threads that process env slices")
    for argument in custom_parameters:
        if ("name" in argument) and ("type" in argument or "action" in argument):
            help_str = argument.get("help", "")
            if "type" in argument:
                if "default" in argument:
                    parser.add_argument(argument["name"], type=argument["type"], default=argument["default"], help=help_str)
                else:
                    parser.add_argument(argument["name"], type=argument["type"], help=help_str)
            else:
                parser.add_argument(argument["name"], action=argument["action"], help=help_str)
        else:
            print("\nERROR: command line argument name, type/action must be defined, argument not added to parser")
            print("supported keys: name, type, default, action, help\n")
    args, unknown_args = parser.parse_known_args()
    if unknown_args:
        print("[aerial_gym_simulator_amd.parse_arguments] Unknown args: ", unknown_args)
    args.sim_device_type, args.compute_device_id = parse_device_str(args.sim_device)
    pipeline = args.pipeline.lower()
    assert pipeline == "cpu" or pipeline in ("gpu", "cuda"), f"Invalid pipeline '{args.pipeline}'. Should be either cpu or gpu."
    args.use_gpu_pipeline = pipeline in ("gpu", "cuda")
    if args.sim_device_type != "cuda" and pipeline == "gpu":
        print("Can't use GPU pipeline with CPU Physics. Changing pipeline to 'CPU'.")
        args.pipeline = "CPU"
        args.use_gpu_pipeline = False
    args.physics_engine = 1 if args.flex else 0  # gymapi.SIM_FLEX / SIM_PHYSX: recorded, never consulted here
    args.use_gpu = args.sim_device_type == "cuda"
    if no_graphics and args.nographics:
        args.headless = True
    if args.slices is None:
        args.slices = args.subscenes
    return args


def update_cfg_from_args(cfg, args):  # utils/helpers.py:82-89
    if cfg is None:
        raise ValueError("cfg is None")
    if args.headless is not None:
        cfg["viewer"]["headless"] = args.headless
    if args.num_envs is not None:
        cfg["env"]["num_envs"] = args.num_envs
    return cfg


def get_args(additional_parameters=None):
    """utils/helpers.py:162-197: --headless, --num_envs, --use_warp on top of parse_arguments, with the reference's
    name alignment (sim_device_id, sim_device = 'cuda:N')."""
    custom_parameters = [
        {"name": "--headless", "type": _strtobool, "default": False, "help": "Force display off at all times"},
        {"name": "--num_envs", "type": int, "default": "64", "help": "Number of environments to create. Overrides config file if provided."},
        {"name": "--use_warp", "type": _strtobool, "default": True, "help": "Use warp for rendering"},
    ]
    args = parse_arguments(description="RL Policy", custom_parameters=custom_parameters + list(additional_parameters or []))
    args.sim_device_id = args.compute_device_id
    args.sim_device = args.sim_device_type
    if args.sim_device == "cuda":
        args.sim_device += f":{args.sim_device_id}"
    return args
