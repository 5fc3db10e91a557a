"""`aerial_gym` -- import-path alias of `aerial_gym_simulator_amd`, so that code written against the reference
(ntnu-arl/aerial_gym_simulator: its rl_training scripts, examples, user task files) imports unchanged:

    from aerial_gym.registry.task_registry import task_registry      # reference rl_games/runner.py:9
    from aerial_gym.utils.helpers import parse_arguments              # reference rl_games/runner.py:10
    from aerial_gym.sim.sim_builder import SimBuilder                 # reference examples/*.py

Every `aerial_gym.x.y` resolves to THE SAME module object as `aerial_gym_simulator_amd.x.y` (one set of registries,
one library handle), through a meta-path alias finder; nothing is imported twice.  The package directory must be on
sys.path (repo root, or `pip install -e .`).  Together with the inert `isaacgym` package next to it this is what
"drop-in for the rl_training scripts" means; INTEGRATION.md lists what is and is not covered."""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import aerial_gym_simulator_amd as _impl

_ALIAS, _REAL = __name__, _impl.__name__


# The reference keeps one file per config class (aerial_gym/config/<group>/<file>.py, some of them under a further
# sub-package: sensor_config/camera_config/...); here a group is ONE module.  `aerial_gym.config.<group>.<file>` therefore
# resolves to a synthetic module that hands out the group module's names -- plus the generic names some reference files use
# for their class (`task_config`, `control`), mapped below to the class that file defines.
_FILE_EXPORTS = {
    "task_config.position_setpoint_task_config": {"task_config": "position_setpoint_task_config"},
    "task_config.navigation_task_config": {"task_config": "navigation_task_config"},
    "task_config.lidar_navigation_task_config": {"task_config": "lidar_navigation_task_config"},
    "controller_config.lee_controller_config": {"control": "lee_controller_config"},
    "controller_config.lee_controller_config_octarotor": {"control": "lee_controller_config_octarotor"},
    "controller_config.magpie_controller_config": {"control": "magpie_controller_config"},
    "controller_config.lmf2_controller_config": {"control": "lmf2_controller_config"},
    "controller_config.fully_actuated_controller_rov": {"control": "fully_actuated_controller_config"},
    "controller_config.no_control_config": {"control": "no_control_config"},
}


# the module paths under aerial_gym/config/ of the reference (files, and the three sensor sub-packages): only these resolve
_REFERENCE_CONFIG_MODULES = frozenset("""
    asset_config.base_asset asset_config.dynamic_env_object_config asset_config.env_asset_config
    asset_config.env_object_config asset_config.lidar_nav_env_config controller_config.fully_actuated_controller_rov
    controller_config.lee_controller_config controller_config.lee_controller_config_octarotor
    controller_config.lmf2_controller_config controller_config.magpie_controller_config
    controller_config.no_control_config env_config.base_env_config env_config.dynamic_environment env_config.empty_env
    env_config.env_config_2ms env_config.env_with_lidar_nav_obstacles env_config.env_with_obstacles
    env_config.forest_env robot_config.base_octarotor_config robot_config.base_quad_config
    robot_config.base_quad_root_link_control_config robot_config.base_random_config robot_config.base_rov_config
    robot_config.lmf1_config robot_config.lmf2_config robot_config.lmf2_radar_config robot_config.magpie_config
    robot_config.morphy_config robot_config.morphy_stiff_config robot_config.snakey5_config
    robot_config.snakey6_config robot_config.snakey_config robot_config.tinyprop_config robot_config.x500_config
    sensor_config.base_sensor_config sensor_config.camera_config.base_depth_camera_config
    sensor_config.camera_config.base_normal_faceID_camera_config sensor_config.camera_config.d455_depth_config
    sensor_config.camera_config.intel_realsense_d455_config sensor_config.camera_config.luxonis_oak_d_config
    sensor_config.camera_config.luxonis_oak_d_pro_w_config sensor_config.camera_config.stereo_camera_config
    sensor_config.imu_config.base_imu_config sensor_config.imu_config.bosch_bmi088_config
    sensor_config.imu_config.vn100_config sensor_config.lidar_config.base_lidar_config
    sensor_config.lidar_config.fake_radar_config sensor_config.lidar_config.os0_128_config
    sensor_config.lidar_config.os0_64_config sensor_config.lidar_config.os1_64_config
    sensor_config.lidar_config.os2_64_config sensor_config.lidar_config.osdome_64_config
    sensor_config.lidar_config.pmd_flexx2_config sensor_config.lidar_config.rslidar_airy_config
    sensor_config.lidar_config.st_vl53l5cx_config sim_config.base_sim_config sim_config.base_sim_headless_config
    sim_config.base_sim_no_gravity_config sim_config.custom_sim_config sim_config.sim_config_2ms
    sim_config.sim_config_4ms task_config.lidar_navigation_task_config task_config.navigation_task_config
    task_config.position_setpoint_task_acceleration_sim2real_config task_config.position_setpoint_task_config
    task_config.position_setpoint_task_config_reconfigurable task_config.position_setpoint_task_lmf2_config
    task_config.position_setpoint_task_morphy_config task_config.position_setpoint_task_sim2real_config
    task_config.position_setpoint_task_sim2real_end_to_end_config
    task_config.position_setpoint_task_sim2real_px4_config task_config.radar_navigation_task_config
    sensor_config.camera_config sensor_config.imu_config sensor_config.lidar_config
""".split())


class _ConfigFileModule(types.ModuleType):
    """`aerial_gym.config.<group>.<file>[...]`: attribute access falls through to the group module"""

    def __getattr__(self, name):
        group, exports = self.__dict__["_agx_group"], self.__dict__["_agx_exports"]
        try:
            return getattr(group, exports.get(name, name))
        except AttributeError:
            raise AttributeError(f"module {self.__name__!r} has no attribute {name!r} (the aerial_gym alias serves the names of "
                                 f"{group.__name__}; the reference's class for it is not part of this package)") from None


class _ConfigGroupModule(types.ModuleType):
    """class of the aliased config group modules: importing `aerial_gym.config.task_config.navigation_task_config` makes the
    import system bind the synthetic per-file module as an attribute of its parent -- which must not replace the CLASS of the
    same name that lives there"""

    def __setattr__(self, name, value):
        if isinstance(value, _ConfigFileModule) and name in self.__dict__:
            return
        super().__setattr__(name, value)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_ALIAS + "."):
            return None
        real = _REAL + fullname[len(_ALIAS):]
        try:
            mod = importlib.import_module(real)
        except ModuleNotFoundError as e:
            if e.name is None or not real.startswith(e.name):  # a genuine missing dependency inside the implementation
                raise
            spec = self._config_file_spec(fullname)
            return spec  # None: the reference has sub-packages this path does not rebuild
        parts = fullname.split(".")
        if len(parts) == 3 and parts[1] == "config" and not hasattr(mod, "__path__"):
            # a config group is one module here and a package of per-file modules in the reference: it has to look like a
            # package BEFORE `import aerial_gym.config.<group>.<file>` asks for its __path__
            mod.__path__ = []
            mod.__class__ = _ConfigGroupModule
        spec = importlib.machinery.ModuleSpec(fullname, self, is_package=hasattr(mod, "__path__"))
        spec._agx_module = mod
        return spec

    def _config_file_spec(self, fullname):
        parts = fullname.split(".")
        if len(parts) < 4 or parts[1] != "config" or ".".join(parts[2:]) not in _REFERENCE_CONFIG_MODULES:
            return None
        try:
            group = importlib.import_module(".".join([_REAL, "config", parts[2]]))
        except ModuleNotFoundError:
            return None
        synth = _ConfigFileModule(fullname)
        synth.__dict__["_agx_group"] = group
        synth.__dict__["_agx_exports"] = _FILE_EXPORTS.get(".".join(parts[2:]), {})
        synth.__path__ = []  # further levels (sensor_config.camera_config.<file>) resolve the same way
        spec = importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        spec._agx_module = synth
        return spec

    def create_module(self, spec):
        mod = spec._agx_module  # the implementation module itself: aliases share state
        spec._agx_saved = (getattr(mod, "__spec__", None), getattr(mod, "__loader__", None), getattr(mod, "__package__", None))
        return mod

    def exec_module(self, module):
        # importlib has just stamped the ALIAS's spec / loader onto the implementation module (_init_module_attrs): put its
        # own back, so that importlib.reload, pkgutil and inspect keep seeing `aerial_gym_simulator_amd.x` as what it is
        if isinstance(module, _ConfigFileModule):
            return
        saved = getattr(module.__spec__, "_agx_saved", None)
        if saved is not None and saved[0] is not None:
            module.__spec__, module.__loader__, module.__package__ = saved


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

AERIAL_GYM_DIRECTORY = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # aerial_gym/__init__.py:4
for _name in ("registry", "config", "control", "robots", "env_manager", "sensors", "sim", "task", "utils", "assets"):
    globals()[_name] = importlib.import_module(_ALIAS + "." + _name)
