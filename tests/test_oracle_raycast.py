"""CPU: known-answer tests that pin the ray-cast oracle geometrically (the reference ships no
vectors for this path: Warp is a third-party dependency), and BVH == brute force."""
import numpy as np
from scene_util import random_box_scene, random_robot_states


def _wall_scene():
    """one 2 x 2 m plate at x = 3 (two triangles), facing -x"""
    a, b, c, d = [3, -1, -1], [3, 1, -1], [3, 1, 1], [3, -1, 1]
    tris = np.array([[a + b + c, a + c + d]], np.float32)  # [1, 2, 9]
    return tris, np.array([[7, 9]], np.int32)


def test_camera_depth_on_axis_aligned_wall(orc):
    tris, seg = _wall_scene()
    W, H = 16, 12
    kinv, cx, cy = orc.camera_kinv(W, H, 60.0)
    # x-forward body convention: sensor quat = q_frame(-90, 0, -90 deg) (warp_sensor.py:104-109)
    frame = orc.quat_from_euler(np.deg2rad(np.array([[-90.0, 0.0, -90.0]], np.float32)))
    pos = np.zeros((1, 1, 3), np.float32)
    quat = frame.reshape(1, 1, 4)
    depth, s = orc.raycast_camera(W, H, kinv, 10.0, cx, cy, "depth", pos, quat, tris, seg)
    rng_, _ = orc.raycast_camera(W, H, kinv, 10.0, cx, cy, "range", pos, quat, tris, seg)
    hit = s[0, 0] != -2
    assert hit.sum() > 20 and (~hit).sum() > 0
    assert np.allclose(depth[0, 0][hit], 3.0, rtol=2e-6)  # depth along the principal axis
    assert np.all(depth[0, 0][~hit] == 1000.0)
    # range = depth / cos(angle): pixel (x, y) has direction (1, -(x-cx)/f, -(y-cy)/f)
    f = (W / 2) / np.tan(np.deg2rad(30.0))
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    expect = 3.0 * np.sqrt(1 + ((xs - W / 2) / f) ** 2 + ((ys - H / 2) / f) ** 2)
    assert np.allclose(rng_[0, 0][hit], expect[hit], rtol=3e-6)
    # which triangle: image row v grows downwards (-z in the body frame); the diagonal a-c splits
    # the plate into z < y (face 0, seg 7) and z > y (face 1, seg 9)
    assert set(np.unique(s[0, 0][hit])) == {7, 9}
    pc, _ = orc.raycast_camera(W, H, kinv, 10.0, cx, cy, "pointcloud_world", pos, quat, tris, seg)
    assert np.allclose(pc[0, 0][hit][:, 0], 3.0, rtol=3e-6)


def test_max_range_and_behind(orc):
    tris, seg = _wall_scene()
    W, H = 8, 8
    kinv, cx, cy = orc.camera_kinv(W, H, 40.0)
    frame = orc.quat_from_euler(np.deg2rad(np.array([[-90.0, 0.0, -90.0]], np.float32)))
    quat = frame.reshape(1, 1, 4)
    d, s = orc.raycast_camera(W, H, kinv, 2.5, cx, cy, "depth", np.zeros((1, 1, 3), np.float32), quat, tris, seg)
    assert np.all(d == 1000.0) and np.all(s == -2)  # beyond far plane
    d, s = orc.raycast_camera(W, H, kinv, 10.0, cx, cy, "depth", np.array([[[5.0, 0, 0]]], np.float32), quat, tris, seg)
    assert np.all(d == 1000.0)  # wall is behind the camera (t >= 0 only)


def test_lidar_table_and_range(orc):
    rv = orc.lidar_ray_table(8, 16, -180, 180, -45, 45)
    assert np.allclose(np.linalg.norm(rv, axis=2), 1.0, atol=1e-6)
    # endpoints (warp_lidar.py:51-59): first column az = +180 deg, first row el = +45 deg
    assert np.allclose(rv[0, 0], [np.cos(np.pi) * np.cos(np.pi / 4), 0.0, np.sin(np.pi / 4)], atol=1e-6)
    assert np.allclose(rv[-1, -1], [np.cos(-np.pi) * np.cos(-np.pi / 4), 0.0, np.sin(-np.pi / 4)], atol=1e-6)
    tris, seg = _wall_scene()
    rv = orc.lidar_ray_table(5, 9, -15, 15, -10, 10)
    quat = np.array([[[0, 0, 0, 1.0]]], np.float32)
    r, s = orc.raycast_lidar(rv, 10.0, "range", np.zeros((1, 1, 3), np.float32), quat, tris, seg)
    hit = s[0, 0] != -2
    assert hit.all()
    assert np.allclose(r[0, 0], 3.0 / rv[..., 0], rtol=3e-6)


def test_bvh_equals_brute_force(orc):
    sc = random_box_scene(3, 40, seed=4)
    tris = orc.scene_transform(sc["tri_local"], sc["tri_asset"], sc["asset_state"])
    st = random_robot_states(3, 1, *sc["bounds"])
    lp = np.tile(np.array([0.1, 0, 0.03], np.float32), (3, 1, 1))
    lq = np.tile(np.array([0, 0, 0, 1], np.float32), (3, 1, 1))
    frame = orc.quat_from_euler(np.deg2rad(np.array([[-90.0, 0.0, -90.0]], np.float32)))[0]
    pos, quat = orc.sensor_pose(st, lp, lq, frame)
    kinv, cx, cy = orc.camera_kinv(64, 48, 87.0)
    a = orc.raycast_camera(64, 48, kinv, 10.0, cx, cy, "depth", pos, quat, tris, sc["tri_seg"], use_bvh=False)
    b = orc.raycast_camera(64, 48, kinv, 10.0, cx, cy, "depth", pos, quat, tris, sc["tri_seg"], use_bvh=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert (a[1] != -2).mean() > 0.5


def test_postprocess_semantics(orc):
    px = np.array([0.05, 0.2, 5.0, 10.0, 10.5, 1000.0], np.float32)
    out = orc.sensor_postprocess(px.copy(), 0.2, 10.0, 10.0, -10.0, True)
    assert np.allclose(out, [-1.0, 0.02, 0.5, 1.0, 1.0, 1.0])
    out = orc.sensor_postprocess(px.copy(), 0.2, 10.0, -1.0, -1.0, False)
    assert np.allclose(out, [-1.0, 0.2, 5.0, 10.0, -1.0, -1.0])  # far value -1 is then < min_range -> near value
