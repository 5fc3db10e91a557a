"""aerial_gym/env_manager/base_env_manager.py"""


class BaseManager:
    def __init__(self, config, device):
        self.cfg = config
        self.device = device

    def reset(self):
        raise NotImplementedError

    def reset_idx(self, env_ids):
        raise NotImplementedError

    def pre_physics_step(self, actions):
        pass

    def step(self, actions):
        pass

    def post_physics_step(self):
        pass

    def init_tensors(self, global_tensor_dict):
        pass
