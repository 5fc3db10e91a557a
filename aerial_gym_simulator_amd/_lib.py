"""ctypes binding of libaerialgym_hip.so -- the ONLY compute back-end of this package.

There is deliberately no CPU / PyTorch fallback: if the library is missing or a tensor
does not live on a HIP device, calls fail loudly (RuntimeError).
"""
import ctypes as C
import os

import torch

from . import _build

MAX_MOTORS = 8
MAX_ACTIONS = 8
MAX_SUBSTEPS = 32
MAX_BODIES = 24
LAUNCH_LEAN, LAUNCH_BODY_WRENCH = 4, 8  # AgxEnvBuffers.launch_flags bits 2 and 3

CTRL_IDS = {
    "none": 0,
    "position": 1,
    "velocity": 2,
    "attitude": 3,
    "rates": 4,
    "acceleration": 5,
    "velocity_steering": 6,
    "fully_actuated": 7,
    "wrench": 8,  # external controller (user class): the kernel takes its output
}


class AgxRobotParams(C.Structure):
    _fields_ = [
        ("num_motors", C.c_int32),
        ("num_actions", C.c_int32),
        ("controller", C.c_int32),
        ("root_link_mode", C.c_int32),
        ("dt", C.c_float),
        ("dt_over_6", C.c_float),
        ("gravity", C.c_float * 3),
        ("mass", C.c_float),
        ("inertia", C.c_float * 9),
        ("inertia_inv", C.c_float * 9),
        ("alloc", C.c_float * (6 * MAX_MOTORS)),
        ("alloc_pinv", C.c_float * (MAX_MOTORS * 6)),
        ("wrench_map", C.c_float * (6 * MAX_MOTORS)),
        ("motor_dir", C.c_float * MAX_MOTORS),
        ("cq", C.c_float),
        ("use_rps", C.c_int32),
        ("use_discrete_approximation", C.c_int32),
        ("integration_rk4", C.c_int32),
        ("min_thrust", C.c_float),
        ("max_thrust", C.c_float),
        ("max_rate", C.c_float),
        ("max_yaw_rate", C.c_float),
        ("lin_drag_linear", C.c_float * 3),
        ("lin_drag_quadratic", C.c_float * 3),
        ("ang_drag_linear", C.c_float * 3),
        ("ang_drag_quadratic", C.c_float * 3),
        ("linear_damping", C.c_float),
        ("angular_damping", C.c_float),
        ("max_linear_velocity", C.c_float),
        ("max_angular_velocity", C.c_float),
        ("collision_radius", C.c_float),
        ("gains_uniform", C.c_float * 12),
        ("tau_inc_uniform", C.c_float),
        ("tau_dec_uniform", C.c_float),
    ]


class AgxEnvBuffers(C.Structure):
    _fields_ = [
        ("state", C.c_void_p),
        ("derived", C.c_void_p),
        ("actions", C.c_void_p),
        ("prev_actions", C.c_void_p),
        ("motor_thrust", C.c_void_p),
        ("motor_kT", C.c_void_p),
        ("motor_tau_inc", C.c_void_p),
        ("motor_tau_dec", C.c_void_p),
        ("gains", C.c_void_p),
        ("wrench_cmd", C.c_void_p),
        ("crashes", C.c_void_p),
        ("truncations", C.c_void_p),
        ("sim_steps", C.c_void_p),
        ("reset_mask", C.c_void_p),
        ("reset_flag", C.c_void_p),
        ("flag_parity", C.c_int32),
        ("episode_count", C.c_void_p),
        ("bounds_min", C.c_void_p),
        ("bounds_max", C.c_void_p),
        ("disturb", C.c_void_p),
        ("disturb_max", C.c_float * 6),
        ("disturb_prob", C.c_float),
        ("step_counter", C.c_int32),
        ("rng_seed", C.c_uint64),
        ("boxes", C.c_void_p),
        ("num_boxes", C.c_int32),
        ("step_rows", C.c_void_p * 2),
        ("step_reward", C.c_void_p),
        ("step_signal", C.c_void_p),
        ("push_delta", C.c_int64 * 7),
        ("push_flags", C.c_void_p * 8),
        ("push_world", C.c_int32),
        ("push_rank", C.c_int32),
        ("push_flag_index", C.c_int32),
        ("push_wait_index", C.c_int32),
        ("push_pub_index", C.c_int32),
        ("push_pub_seq", C.c_uint32),
        ("push_seq", C.c_uint32),
        ("push_wait_seq", C.c_uint32),
        ("push_timed_out", C.c_void_p),
        ("push_base", C.c_void_p),
        ("push_slice_bytes", C.c_int64),
        ("push_slots", C.c_int32),
        ("body_force", C.c_void_p),
        ("step_counter_dev", C.c_void_p),
        ("env_index_base", C.c_int32),
        ("launch_flags", C.c_int32),
    ]


class AgxImuArgs(C.Structure):
    _fields_ = [
        ("bias_std", C.c_float * 6),
        ("noise_std", C.c_float * 6),
        ("max_value", C.c_float * 6),
        ("max_bias_init", C.c_float * 6),
        ("min_rot", C.c_float * 3),
        ("max_rot", C.c_float * 3),
        ("g_world", C.c_float * 3),
        ("sqrt_dt", C.c_float),
        ("mass", C.c_float),
        ("world_frame", C.c_int32),
        ("enable_noise", C.c_int32),
        ("enable_bias", C.c_int32),
    ]


class AgxTaskArgs(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("episode_len", C.c_int32),
        ("reset_on_collision", C.c_int32),
        ("curriculum_progress", C.c_float),
        ("target", C.c_void_p),
        ("reward", C.c_void_p),
        ("pos_err", C.c_void_p),
        ("prev_pos_err", C.c_void_p),
        ("rp", C.c_float * 18),
        ("successes", C.c_void_p),
        ("timeouts", C.c_void_p),
        ("counters", C.c_void_p),
        ("success_radius", C.c_float),
        ("reserved", C.c_int32),
    ]


TASK_NONE, TASK_POSITION, TASK_NAVIGATION = 0, 1, 2


class AgxNavRobotSideArgs(C.Structure):
    _fields_ = [
        ("num_sensors", C.c_int32),
        ("randomize_mount", C.c_int32),
        ("mount_t_min", C.c_float * 3),
        ("mount_t_max", C.c_float * 3),
        ("mount_r_min", C.c_float * 3),
        ("mount_r_max", C.c_float * 3),
        ("local_pos", C.c_void_p),
        ("local_quat", C.c_void_p),
        ("frame_quat", C.c_float * 4),
        ("sensor_pos", C.c_void_p),
        ("sensor_quat", C.c_void_p),
        ("reset_target", C.c_int32),
        ("num_actions", C.c_int32),
        ("zero_prev_actions", C.c_int32),
        ("pad_", C.c_int32),
        ("target_ratio_min", C.c_float * 3),
        ("target_ratio_max", C.c_float * 3),
        ("target", C.c_void_p),
        ("target_yaw", C.c_void_p),
    ]


class AgxResetArgs(C.Structure):
    _fields_ = [
        ("u_bounds_lo", C.c_void_p),
        ("u_bounds_hi", C.c_void_p),
        ("u_state", C.c_void_p),
        ("u_gains", C.c_void_p),
        ("u_tau_inc", C.c_void_p),
        ("u_tau_dec", C.c_void_p),
        ("u_thrust", C.c_void_p),
        ("u_kT", C.c_void_p),
        ("lower_bound_min", C.c_float * 3),
        ("lower_bound_max", C.c_float * 3),
        ("upper_bound_min", C.c_float * 3),
        ("upper_bound_max", C.c_float * 3),
        ("min_state", C.c_float * 13),
        ("max_state", C.c_float * 13),
        ("gains_min", C.c_float * 12),
        ("gains_max", C.c_float * 12),
        ("tau_inc_min", C.c_float),
        ("tau_inc_max", C.c_float),
        ("tau_dec_min", C.c_float),
        ("tau_dec_max", C.c_float),
        ("kT_min", C.c_float),
        ("kT_max", C.c_float),
        ("randomize_gains", C.c_int32),
        ("seed", C.c_uint64),
    ]


class AgxRangeLimits(C.Structure):
    _fields_ = [("min_range", C.c_float), ("max_range", C.c_float), ("far_oor", C.c_float), ("near_oor", C.c_float),
                ("normalize", C.c_int32)]


class AgxPositionStepPlan(C.Structure):
    _fields_ = [
        ("params", C.POINTER(AgxRobotParams)),
        ("buf", C.POINTER(AgxEnvBuffers)),
        ("task", C.POINTER(AgxTaskArgs)),
        ("reset", C.POINTER(AgxResetArgs)),
        ("target", C.c_void_p),
        ("obs", C.c_void_p),
        ("num_envs", C.c_int32),
        ("k_substeps", C.c_int32),
    ]


MAX_UNIFORM_SEGMENTS = 8


class AgxStrictStepPlan(C.Structure):
    _fields_ = [("plan", C.POINTER(AgxPositionStepPlan)), ("host_word", C.POINTER(C.c_uint32)), ("count", C.c_int32), ("timeout_ms", C.c_int32),
                ("out", C.c_void_p * MAX_UNIFORM_SEGMENTS), ("numel", C.c_int64 * MAX_UNIFORM_SEGMENTS), ("seed", C.c_uint64),
                ("offset", C.c_uint64), ("sm_count", C.c_int32), ("max_threads_per_sm", C.c_int32)]


class AgxRobotStepArgs(C.Structure):
    _fields_ = [("force", C.c_void_p), ("torque", C.c_void_p), ("num_bodies", C.c_int32), ("substep", C.c_int32),
                ("body_of_motor", C.c_int32 * MAX_MOTORS)]


class AgxLinkFrames(C.Structure):
    _fields_ = [("num_bodies", C.c_int32), ("reserved", C.c_int32), ("rot", (C.c_float * 9) * MAX_BODIES), ("pos", (C.c_float * 3) * MAX_BODIES)]


ABI_VERSION = 12  # AGX_ABI_VERSION of include/aerial_gym_hip.h these mirrors were written against
_P = C.c_void_p
_SIGNATURES = {
    "agx_last_error": (C.c_char_p, []),
    "agx_abi_version": (C.c_int, []),
    "agx_build_id": (C.c_char_p, []),
    "agx_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "agx_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "agx_math_eval": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P]),
    "agx_copy_f4": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "agx_dynamics_substeps": (C.c_int, [C.POINTER(AgxRobotParams), C.POINTER(AgxEnvBuffers), C.c_int, _P, C.c_int, _P]),
    "agx_env_step": (C.c_int, [C.POINTER(AgxRobotParams), C.POINTER(AgxEnvBuffers), C.c_int, _P, C.c_int, C.POINTER(AgxTaskArgs), _P]),
    "agx_update_states": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, _P]),
    "agx_collide_spheres_boxes": (C.c_int, [C.POINTER(AgxRobotParams), C.POINTER(AgxEnvBuffers), C.c_int, _P]),
    "agx_controller_wrench": (C.c_int, [C.POINTER(AgxRobotParams), C.POINTER(AgxEnvBuffers), C.c_int, _P, _P]),
    "agx_robot_step": (C.c_int, [C.POINTER(AgxRobotParams), C.POINTER(AgxEnvBuffers), C.c_int, _P, C.POINTER(AgxRobotStepArgs), _P]),
    "agx_net_body_wrench": (C.c_int, [C.c_int, C.POINTER(AgxLinkFrames), _P, _P, _P, _P]),
    "agx_reward_position": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    "agx_obs_position": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, _P, _P, _P]),
    "agx_reward_navigation": (
        C.c_int,
        [C.POINTER(AgxEnvBuffers), C.c_int, _P, C.POINTER(C.c_float), C.c_float, _P, _P, C.c_int, C.c_int, _P, _P],
    ),
    "agx_obs_navigation": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, _P, _P, _P]),
    "agx_env_step_kernel": (C.c_int, [C.POINTER(AgxRobotParams), C.POINTER(AgxEnvBuffers), C.c_int, C.c_int, C.POINTER(AgxTaskArgs),
                                      C.c_char_p, C.c_int]),
    "agx_raycast_kernel": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "agx_reset_masked": (C.c_int, [C.POINTER(AgxRobotParams), C.POINTER(AgxEnvBuffers), C.c_int, C.POINTER(AgxResetArgs), _P]),
    "agx_nav_robot_side": (C.c_int, [C.POINTER(AgxRobotParams), C.POINTER(AgxEnvBuffers), C.c_int, C.POINTER(AgxResetArgs),
                                     C.POINTER(AgxNavRobotSideArgs), _P]),
    "agx_post_step_position": (C.c_int, [C.POINTER(AgxRobotParams), C.POINTER(AgxEnvBuffers), C.c_int, C.POINTER(AgxResetArgs), _P, _P, _P]),
    "agx_position_task_step": (C.c_int, [C.POINTER(AgxPositionStepPlan), _P, _P]),
    "agx_torch_uniform_fill": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                         C.POINTER(C.c_uint64), _P]),
    "agx_host_word_create": (C.c_int, [C.POINTER(C.POINTER(C.c_uint32))]),
    "agx_host_word_destroy": (C.c_int, [C.POINTER(C.c_uint32)]),
    "agx_position_task_step_strict": (C.c_int, [C.POINTER(AgxStrictStepPlan), _P, C.POINTER(C.c_int), C.POINTER(C.c_uint64), _P]),
    "agx_reset_assets": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, C.c_int, C.POINTER(AgxResetArgs), _P, _P, _P, _P, _P, C.c_int,
                                   C.c_int, _P, _P]),
    "agx_scene_transform": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "agx_bvh_nodes_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "agx_prims_from_assets": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "agx_assets_integrate": (C.c_int, [C.c_int, C.c_int, _P, _P, C.c_float, C.c_int, _P]),
    "agx_exchange_unique_id": (C.c_int, [C.c_char_p, _P, C.c_int]),
    "agx_exchange_create": (C.c_int, [C.c_char_p, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "agx_exchange_post": (C.c_int, [_P, C.c_int, _P, _P, C.c_size_t, _P, C.c_uint32, _P]),
    "agx_exchange_probe": (C.c_int, [_P, _P]),
    "agx_exchange_wait": (C.c_int, [_P, C.c_int, _P]),
    "agx_exchange_step": (C.c_int, [_P, C.c_int, _P, _P, C.c_size_t, _P, C.c_uint32, C.c_int, _P]),
    "agx_exchange_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "agx_exchange_destroy": (C.c_int, [_P]),
    "agx_exchange_create_push": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "agx_exchange_push_export": (C.c_int, [_P, _P, C.c_int]),
    "agx_exchange_push_connect": (C.c_int, [_P, _P, C.c_int]),
    "agx_exchange_push_buffer": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "agx_exchange_push_peers": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "agx_exchange_push_wait_seq": (C.c_int, [_P, C.c_uint32, _P]),
    "agx_exchange_check": (C.c_int, [_P]),
    "agx_exchange_push_selftest": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), _P]),
    "agx_action_transform": (C.c_int, [C.c_int, C.c_int, _P, _P, _P]),
    "agx_push_advance": (C.c_int, [C.POINTER(AgxEnvBuffers)]),
    "agx_bvh_build": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "agx_boxes_from_assets": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "agx_scene_refresh": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "agx_scene_reset_refresh": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, C.c_int, C.c_int, C.POINTER(AgxResetArgs), _P, _P, C.c_int, C.c_int, _P,
                                          _P, _P, _P, C.c_int, _P, _P, _P, _P, _P]),
    "agx_sensor_pose": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "agx_raycast_camera": (
        C.c_int,
        [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P,
         C.c_int, _P, _P, C.POINTER(AgxRangeLimits), _P],
    ),
    "agx_raycast_stereo_camera": (
        C.c_int,
        [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, _P, _P, _P,
         _P, _P, C.c_int, _P, _P, C.POINTER(AgxRangeLimits), _P],
    ),
    "agx_raycast_lidar": (
        C.c_int,
        [C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_float, C.c_int, _P, _P, _P, _P, _P, C.c_int, _P, _P, C.POINTER(AgxRangeLimits), _P],
    ),
    "agx_sensor_postprocess": (
        C.c_int,
        [C.c_size_t, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
         C.c_float, C.c_int, _P],
    ),
    "agx_sensor_postprocess_points": (
        C.c_int,
        [C.c_size_t, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
         C.c_float, C.c_int, C.c_int, _P],
    ),
    "agx_image_min": (C.c_int, [C.c_int, C.c_int, _P, _P, _P]),
    "agx_step_counter_advance": (C.c_int, [C.POINTER(AgxEnvBuffers), _P]),
    "agx_reset_set": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, C.c_int, _P]),
    "agx_nav_bookkeeping": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, _P, C.c_float, _P, _P, _P, _P]),
    "agx_nav_target_reset": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, _P,
                                       C.c_int, _P]),
    "agx_sensor_mount_reset": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                         C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, _P, _P, _P]),
    "agx_imu_update": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, C.c_int, C.POINTER(AgxImuArgs), _P, _P, _P, _P, _P, _P]),
    "agx_imu_reset": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, C.POINTER(AgxImuArgs), _P, _P, _P, _P, _P]),
    "agx_lidar_image_obs": (
        C.c_int,
        [C.POINTER(AgxEnvBuffers), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P],
    ),
    "agx_reward_lidar_navigation": (
        C.c_int,
        [C.POINTER(AgxEnvBuffers), C.c_int, _P, _P, _P, _P, _P, C.POINTER(C.c_float), C.c_float, _P, _P, C.c_int, C.c_int, _P, _P],
    ),
    "agx_obs_lidar_navigation": (C.c_int, [C.POINTER(AgxEnvBuffers), C.c_int, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())

_lib = None


def library_path():
    return _build.LIB_PATH


def load():
    """Load the HIP library; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("AGX_LIB_PATH", _build.LIB_PATH)  # experimental builds (profiles/variants.py)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found. Build it with `python -m aerial_gym_simulator_amd._build` "
            "(needs hipcc); this package has no CPU or PyTorch fallback."
        )
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{path} does not export {name}: stale build? ({e})") from None
        fn.restype = res
        fn.argtypes = args
    if lib.agx_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libaerialgym_hip.so ABI version {lib.agx_abi_version()} != {ABI_VERSION}: stale build?")
    _lib = lib
    return lib


def build_id():
    """identity of the LOADED binary (hash of the sources + flags it was compiled from, `agx_build_id`)"""
    return load().agx_build_id().decode()


def binary_matches_sources():
    """True when the loaded library was built from the kernel sources lying next to it (counter files under profiles/ are
    stamped with `build_id()`; a stale library shipped with newer sources shows up here, not in file times)"""
    return build_id() == _build.source_hash()


def set_option(name, value):
    """agx_set_option: process-wide tuning / A-B switches of the library ("env_step_quad", "ray_split"; include/aerial_gym_hip.h)"""
    check(load().agx_set_option(name.encode(), int(value)), "agx_set_option")


def get_option(name):
    v = C.c_int(0)
    check(load().agx_get_option(name.encode(), C.byref(v)), "agx_get_option")
    return v.value


class option:
    """`with _lib.option("env_step_quad", 0): ...` -- the previous value comes back on exit"""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def check(code, what=""):
    if code != 0:
        msg = load().agx_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what or 'agx call'} failed ({code}): {msg}")


def dptr(t):
    """Raw device pointer of a tensor that MUST live on a HIP device and be dense."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(
            "aerial_gym_simulator_amd kernels need tensors on a HIP device (got %s): there is no CPU fallback" % t.device
        )
    if not t.is_contiguous():
        raise RuntimeError("tensor handed to the HIP library must be contiguous")
    return t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (device index) -> hipStream_t as int, ~0.2 us
_DEVICE_INDEX = {}


def current_stream(device):
    """torch's current stream on `device` as a hipStream_t.  torch.cuda.current_stream(device).cuda_stream builds a Stream
    object under a device guard (2.3 us, a fifth of the host's share of a position-task step:
    profiles/r02_host_cost_probe.json); the raw lookup returns the same handle."""
    if _RAW_STREAM is None:
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    idx = _DEVICE_INDEX.get(device)
    if idx is None:
        d = torch.device(device)
        if d.index is None:  # "cuda": whatever device is current at the time of the call
            return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
        idx = _DEVICE_INDEX[device] = d.index
    return C.c_void_p(_RAW_STREAM(idx))
