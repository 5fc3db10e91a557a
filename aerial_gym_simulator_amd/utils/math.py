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


# ---- the rest of the reference's helper names (aerial_gym/utils/math.py), for user tasks / controllers that do
# `from aerial_gym.utils.math import *`: textbook quaternion / sampling helpers, same argument order and conventions
# (xyzw quaternions, angles in radians, batch in the leading dimension) --------------------------------------------
def compute_vee_map(skew_matrix):  # :35-42  vee of a batch of skew-symmetric matrices
    return torch.stack([-skew_matrix[:, 1, 2], skew_matrix[:, 0, 2], -skew_matrix[:, 0, 1]], dim=1)


def torch_rand_float(lower, upper, shape, device):  # :207-210
    return lower + (upper - lower) * torch.rand(*shape, device=device)


def torch_rand_float_vec(lower, upper, shape, device):  # :46-49
    return lower + torch.rand(*shape, device=device) * (upper - lower)


def torch_random_dir_2(shape, device):  # :213-217  unit vectors in the plane
    ang = torch_rand_float(-torch.pi, torch.pi, shape, device).squeeze(-1)
    return torch.stack([ang.cos(), ang.sin()], dim=-1)


def tensor_clamp(t, min_t, max_t):  # :220-222
    return torch.maximum(torch.minimum(t, max_t), min_t)


def scale(x, lower, upper):  # [-1, 1] -> [lower, upper]
    return lower + 0.5 * (x + 1.0) * (upper - lower)


def unscale(x, lower, upper):  # [lower, upper] -> [-1, 1]
    return (2.0 * x - upper - lower) / (upper - lower)


unscale_np = unscale


def to_torch(x, dtype=torch.float, device="cuda:0", requires_grad=False):
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)


def normalize(x, eps: float = 1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps).unsqueeze(-1)


def quat_inverse(a):  # unit quaternions
    return quat_conjugate(a)


def quat_unit(a):
    return normalize(a)


def quat_from_angle_axis(angle, axis):
    half = (angle / 2).unsqueeze(-1)
    return quat_unit(torch.cat([normalize(axis) * half.sin(), half.cos()], dim=-1))


def quat_axis(q, axis=0):  # :69-73  the rotated basis vector e_axis
    e = torch.zeros(q.shape[0], 3, device=q.device, dtype=q.dtype)
    e[:, axis] = 1
    return quat_rotate(q, e)


def quat_to_rotation_matrix(a):  # :267-293  [N, 4] xyzw -> [N, 3, 3]
    x, y, z, w = a.reshape(-1, 4).unbind(-1)
    xx, yy, zz, xy, xz, yz, xw, yw, zw = x * x, y * y, z * z, x * y, x * z, y * z, x * w, y * w, z * w
    rows = [1 - 2 * (yy + zz), 2 * (xy - zw), 2 * (xz + yw), 2 * (xy + zw), 1 - 2 * (xx + zz), 2 * (yz - xw), 2 * (xz - yw),
            2 * (yz + xw), 1 - 2 * (xx + yy)]
    return torch.stack(rows, dim=-1).view(-1, 3, 3)


def copysign(a, b):  # :93-96  |a| with the sign of b (0 where b is 0, like torch.sign)
    return torch.abs(torch.as_tensor(a, device=b.device, dtype=torch.float)) * torch.sign(b)


def get_euler_xyz_tensor(q):  # :124-146  roll, pitch, yaw in [0, 2 pi)
    x, y, z, w = q.unbind(-1)
    roll = torch.atan2(2.0 * (w * x + y * z), w * w - x * x - y * y + z * z)
    sinp = 2.0 * (w * y - z * x)
    pitch = torch.where(sinp.abs() >= 1, copysign(torch.pi / 2.0, sinp), torch.asin(sinp))
    yaw = torch.atan2(2.0 * (w * z + x * y), w * w + x * x - y * y - z * z)
    return torch.stack([roll % (2 * torch.pi), pitch % (2 * torch.pi), yaw % (2 * torch.pi)], dim=-1)


def get_euler_xyz(q):  # :100-121
    e = get_euler_xyz_tensor(q)
    return e[..., 0], e[..., 1], e[..., 2]


def vehicle_frame_quat_from_quat(body_quat):  # :176-180  yaw-only copy of an attitude
    e = get_euler_xyz_tensor(body_quat)
    return quat_from_euler_xyz_tensor(e * torch.tensor([0.0, 0.0, 1.0], device=e.device))


def normalize_angle(x):
    return torch.atan2(torch.sin(x), torch.cos(x))


def tf_inverse(q, t):
    qi = quat_conjugate(q)
    return qi, -quat_apply(qi, t)


def tf_vector(q, v):
    return quat_apply(q, v)


def tf_combine(q1, t1, q2, t2):
    return quat_mul(q1, q2), quat_apply(q1, t2) + t1


def get_basis_vector(q, v):
    return quat_rotate(q, v)


def pd_control(pos_error, vel_error, stiffness, damping):
    return stiffness * pos_error + damping * vel_error
