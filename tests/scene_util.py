"""Test helper: synthetic box scenes in numpy (reference layout) for oracle and kernels."""
import numpy as np

BOX_VERTS = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]], np.float32) - 0.5
BOX_FACES = np.array([[1, 3, 0], [4, 1, 0], [0, 3, 2], [2, 4, 0], [1, 7, 3], [5, 1, 4], [5, 7, 1], [3, 7, 2], [6, 4, 2], [2, 7, 6],
                      [6, 5, 4], [7, 5, 6]])


def quat_from_euler(e):
    r, p, y = e[..., 0] * 0.5, e[..., 1] * 0.5, e[..., 2] * 0.5
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.stack([cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp,
                     cy * cr * cp + sy * sr * sp], axis=-1).astype(np.float32)


def random_box_scene(n, k_boxes, seed=0, bounds=((-2, -4, -3), (10, 4, 3)), walls=True):
    """Returns dict(tri_local [N,T,9], tri_asset [T], tri_seg [N,T], asset_state [N,K,13], half [N,K,3])."""
    rng = np.random.default_rng(seed)
    lo, hi = np.array(bounds[0], np.float32), np.array(bounds[1], np.float32)
    sizes = rng.uniform(0.1, 1.2, (n, k_boxes, 3)).astype(np.float32)
    pos = rng.uniform(lo, hi, (n, k_boxes, 3)).astype(np.float32)
    eul = np.zeros((n, k_boxes, 3), np.float32)
    eul[..., 2] = rng.uniform(-np.pi, np.pi, (n, k_boxes))
    if walls:
        wsz = np.array([[20, 0.2, 20], [20, 0.2, 20], [0.2, 20, 20], [0.2, 20, 20], [20, 20, 0.2], [20, 20, 0.2]], np.float32)
        mid = (lo + hi) / 2
        wpos = np.array([[mid[0], hi[1], mid[2]], [mid[0], lo[1], mid[2]], [lo[0], mid[1], mid[2]], [hi[0], mid[1], mid[2]],
                         [mid[0], mid[1], lo[2]], [mid[0], mid[1], hi[2]]], np.float32)
        sizes = np.concatenate([np.tile(wsz, (n, 1, 1)), sizes], axis=1)
        pos = np.concatenate([np.tile(wpos, (n, 1, 1)), pos], axis=1)
        eul = np.concatenate([np.zeros((n, 6, 3), np.float32), eul], axis=1)
    K = sizes.shape[1]
    tri = BOX_VERTS[BOX_FACES]  # [12,3,3]
    tri_local = (tri[None, None] * sizes[:, :, None, None, :]).reshape(n, 12 * K, 9).astype(np.float32)
    tri_asset = np.repeat(np.arange(K, dtype=np.int32), 12)
    seg = (100 + np.arange(n * K).reshape(n, K)).astype(np.int32)
    if walls:
        seg[:, :6] = np.array([11, 12, 10, 9, 13, 14], np.int32)
    state = np.zeros((n, K, 13), np.float32)
    state[..., 0:3] = pos
    state[..., 3:7] = quat_from_euler(eul)
    return dict(tri_local=np.ascontiguousarray(tri_local), tri_asset=tri_asset, tri_seg=np.ascontiguousarray(np.repeat(seg, 12, axis=1)),
                asset_state=state, half=np.ascontiguousarray(sizes * 0.5), bounds=(lo, hi))


def random_robot_states(n, seed, lo, hi, tilt=0.4):
    rng = np.random.default_rng(seed)
    s = np.zeros((n, 13), np.float32)
    s[:, 0:3] = rng.uniform(lo + 0.5, hi - 0.5, (n, 3))
    e = rng.uniform(-1, 1, (n, 3)) * np.array([tilt, tilt, np.pi])
    s[:, 3:7] = quat_from_euler(e.astype(np.float32))
    return s
