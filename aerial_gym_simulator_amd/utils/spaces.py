"""Minimal gym-style spaces used only when neither `gymnasium` nor `gym` is importable
(the reference imports both: position_setpoint_task.py:10-11)."""
import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box, Dict  # type: ignore
except Exception:  # noqa: BLE001
    try:
        from gym.spaces import Box, Dict  # type: ignore
    except Exception:  # noqa: BLE001

        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.shape = tuple(shape) if shape is not None else np.shape(low)
                self.dtype = np.dtype(dtype)
                self.low = np.full(self.shape, low, dtype=self.dtype)
                self.high = np.full(self.shape, high, dtype=self.dtype)

            def sample(self):
                return np.random.uniform(self.low, self.high).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def __repr__(self):
                return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

        class Dict:
            def __init__(self, spaces):
                self.spaces = dict(spaces)

            def __getitem__(self, key):
                return self.spaces[key]

            def keys(self):
                return self.spaces.keys()

            def __repr__(self):
                return f"Dict({self.spaces})"
