"""Host-side (torch) helpers.  Only set-up / reset glue runs through these; the per-step
math lives in csrc/agx_device_math.h.  Same names and argument meaning as
aerial_gym/utils/math.py so reference-style user code keeps working."""
import torch


def torch_rand_float_tensor(lower, upper):
    return (upper - lower) * torch.rand_like(upper) + lower


def torch_interpolate_ratio(min, max, ratio):  # noqa: A002  (reference argument names)
    return min + (max - min) * ratio


def quat_from_euler_xyz(roll, pitch, yaw):
    hr, hp, hy = roll * 0.5, pitch * 0.5, yaw * 0.5
    cr, sr, cp, sp, cy, sy = hr.cos(), hr.sin(), hp.cos(), hp.sin(), hy.cos(), hy.sin()
    return torch.stack(
        [cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp, cy * cr * cp + sy * sr * sp],
        dim=-1,
    )


def quat_from_euler_xyz_tensor(euler_xyz_tensor):
    return quat_from_euler_xyz(euler_xyz_tensor[..., 0], euler_xyz_tensor[..., 1], euler_xyz_tensor[..., 2])


def quat_conjugate(a):
    return torch.cat((-a[..., :3], a[..., 3:]), dim=-1)


def quat_mul(a, b):
    x1, y1, z1, w1 = a.unbind(-1)
    x2, y2, z2, w2 = b.unbind(-1)
    return torch.stack(
        [
            w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
            w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
            w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
        ],
        dim=-1,
    )


def quat_apply(a, b):
    xyz = a[..., :3]
    t = torch.cross(xyz, b, dim=-1) * 2
    return b + a[..., 3:] * t + torch.cross(xyz, t, dim=-1)


def quat_rotate(q, v):
    return quat_apply(q, v)


def quat_rotate_inverse(q, v):
    return quat_apply(quat_conjugate(q), v)


def quat_apply_inverse(a, b):
    return quat_apply(quat_conjugate(a), b)


def tf_apply(q, t, v):
    return quat_apply(q, v) + t


def ssa(a):
    return torch.remainder(a + torch.pi, 2 * torch.pi) - torch.pi


def exponential_reward_function(magnitude, base_width, value):
    return magnitude * torch.exp(-(value * value) / base_width)


def exponential_penalty_function(magnitude, base_width, value):
    return magnitude * (torch.exp(-(value * value) / base_width) - 1.0)
