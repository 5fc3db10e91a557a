"""Host mirror of aerial_gym/control/control_allocation.py.  The allocation products
(u_ref = pinv(A) w, per-motor force/torque, net body wrench) are evaluated inside
agx_dynamics_substeps with the matrices as kernel constants."""
import torch

from ..utils.logging import CustomLogger
from .motor_model import MotorModel

logger = CustomLogger("control_allocation")


class ControlAllocator:
    def __init__(self, num_envs, dt, config, device, random_source=None):
        self.num_envs, self.dt, self.cfg, self.device = num_envs, dt, config, device
        self.force_application_level = config.force_application_level
        A = torch.tensor(config.allocation_matrix, dtype=torch.float32)
        if A.shape[1] != config.num_motors or A.shape[0] != 6:
            raise ValueError("Allocation matrix must have 6 rows and num_motors columns.")
        rank = int(torch.linalg.matrix_rank(A))
        if rank < 6:
            logger.warning(f"allocation matrix is not full rank. Rank: {rank}")
        self.force_torque_allocation_matrix = A.to(device)
        self.inv_force_torque_allocation_matrix = torch.linalg.pinv(A).to(device)
        self.motor_directions = torch.tensor(config.motor_directions, device=device)
        self.motor_model = MotorModel(num_envs, config.num_motors, dt, config.motor_model_config, device, random_source)
