"""aerial_gym/task/base_task.py:10-54"""
import os
import random
import time
from abc import ABC, abstractmethod

import numpy as np
import torch


class BaseTask(ABC):
    def __init__(self, task_config):
        self.task_config = task_config
        self.action_space = self.observation_space = None
        self.reward_range = self.metadata = self.spec = None
        seed = task_config.seed
        if seed == -1:
            seed = time.time_ns() % (2**32)
        self.seed(seed)

    def seed(self, seed):
        if seed is None or seed < 0:
            seed = time.time_ns() % (2**32)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
        os.environ["PYTHONHASHSEED"] = str(seed)
        random.seed(seed)

    @abstractmethod
    def render(self, mode="human"):
        raise NotImplementedError

    @abstractmethod
    def reset(self):
        raise NotImplementedError

    @abstractmethod
    def reset_idx(self, env_ids):
        raise NotImplementedError

    @abstractmethod
    def step(self, action):
        raise NotImplementedError

    @abstractmethod
    def close(self):
        raise NotImplementedError
