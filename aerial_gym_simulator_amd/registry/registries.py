"""Name -> class/config registries with the reference's method names
(aerial_gym/registry/{task,robot,controller,env,sim}_registry.py)."""


class _Registry:
    kind = "item"

    def __init__(self):
        self._classes = {}
        self._configs = {}

    def _lookup(self, table, name):
        if name not in table:
            raise ValueError(f"{self.kind} {name} not found in {self.kind} registry. Available: {sorted(table)}")
        return table[name]


class TaskRegistry(_Registry):
    kind = "task"

    def register_task(self, task_name, task_class, task_config):
        self._classes[task_name] = task_class
        self._configs[task_name] = task_config

    def get_task_class(self, task_name):
        return self._lookup(self._classes, task_name)

    def get_task_config(self, task_name):
        return self._lookup(self._configs, task_name)

    def get_task_names(self):
        return list(self._classes)

    def get_task_classes(self):
        return list(self._classes.values())

    def get_task_configs(self):
        return list(self._configs.values())

    def make_task(self, task_name, seed=None, num_envs=None, headless=None, use_warp=None):
        cls, cfg = self.get_task_class(task_name), self.get_task_config(task_name)
        return cls(cfg, seed=seed, num_envs=num_envs, headless=headless, use_warp=use_warp)


class ControllerRegistry(_Registry):
    kind = "controller"

    def register_controller(self, controller_name, controller_class, controller_config):
        self._classes[controller_name] = controller_class
        self._configs[controller_name] = controller_config

    def get_controller_class(self, controller_name):
        return self._lookup(self._classes, controller_name)

    def get_controller_names(self):
        return self._classes.keys()

    def get_controller_config(self, controller_name):
        return self._lookup(self._configs, controller_name)

    def make_controller(self, controller_name, num_envs, device, mode="robot"):
        cls, cfg = self.get_controller_class(controller_name), self.get_controller_config(controller_name)
        return cls(cfg, num_envs, device), cfg


class RobotRegistry(_Registry):
    kind = "robot"

    def register(self, robot_name, robot_class, robot_config):
        self._classes[robot_name] = robot_class
        self._configs[robot_name] = robot_config

    def get_robot_class(self, robot_name):
        return self._lookup(self._classes, robot_name)

    def get_robot_config(self, robot_name):
        return self._lookup(self._configs, robot_name)

    def get_robot_names(self):
        return self._classes.keys()

    def make_robot(self, robot_name, controller_name, env_config, device):
        cls, cfg = self.get_robot_class(robot_name), self.get_robot_config(robot_name)
        return cls(cfg, controller_name, env_config, device), cfg


class EnvConfigRegistry(_Registry):
    kind = "env"

    def register(self, env_name, env_config):
        self._configs[env_name] = env_config

    def get_env_config(self, env_name):
        return self._lookup(self._configs, env_name)

    def get_env_names(self):
        return self._configs.keys()

    def make_env(self, env_name):
        return self.get_env_config(env_name)


class SimConfigRegistry(_Registry):
    kind = "sim"

    def register(self, sim_name, sim_config):
        self._configs[sim_name] = sim_config

    def get_sim_config(self, sim_name):
        return self._lookup(self._configs, sim_name)

    def get_sim_names(self):
        return self._configs.keys()

    def make_sim(self, sim_name):
        return self.get_sim_config(sim_name)


task_registry = TaskRegistry()
controller_registry = ControllerRegistry()
robot_registry = RobotRegistry()
env_config_registry = EnvConfigRegistry()
sim_config_registry = SimConfigRegistry()
