"""CPU: known-answer tests that pin the ray-cast oracle geometrically (the reference ships no
vectors for this path: Warp is a third-party dependency), and BVH == brute force."""
import numpy as np
import pytest
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


def _x_forward_camera(orc, W, H, hfov):
    kinv, cx, cy = orc.camera_kinv(W, H, hfov)
    frame = orc.quat_from_euler(np.deg2rad(np.array([[-90.0, 0.0, -90.0]], np.float32)))
    return kinv, cx, cy, np.zeros((1, 1, 3), np.float32), frame.reshape(1, 1, 4)


def test_normal_and_face_id_sensors(orc):
    """warp_camera_kernels.py:70-121 / warp_lidar_kernels.py:90-126: geometric normal of the hit
    face (cross(b-a, c-a) normalised: +x for this plate), zero vector and face -1 on a miss."""
    tris, seg = _wall_scene()
    W, H = 16, 12
    kinv, cx, cy, pos, quat = _x_forward_camera(orc, W, H, 60.0)
    _, s = orc.raycast_camera(W, H, kinv, 10.0, cx, cy, "pointcloud", pos, quat, tris, seg)  # same ray generation (:100-103)
    hit = s[0, 0] != -2
    nw, face = orc.raycast_camera(W, H, kinv, 10.0, cx, cy, "normal_world", pos, quat, tris, seg)
    assert np.array_equal(nw[0, 0][hit], np.tile(np.float32([1, 0, 0]), (hit.sum(), 1)))
    assert np.all(nw[0, 0][~hit] == 0.0) and np.all(face[0, 0][~hit] == -1)
    assert np.array_equal(face[0, 0][hit], np.where(s[0, 0][hit] == 7, 0, 1))  # face index, not semantic id
    # camera frame: (n . rd_p, n . (rd_p x e_z), n . (rd_p x e_y)) with rd_p = +x -> (1, 0, 0)
    nc, _ = orc.raycast_camera(W, H, kinv, 10.0, cx, cy, "normal", pos, quat, tris, seg)
    assert np.allclose(nc[0, 0][hit], [1, 0, 0], atol=1e-6) and np.all(nc[0, 0][~hit] == 0.0)
    # LiDAR yawed by +90 deg looks along +y: the plate is seen by the rays pointing to sensor -y ... use a yaw of -30 deg
    yaw = np.deg2rad(-30.0)
    q = np.array([[[0, 0, np.sin(yaw / 2), np.cos(yaw / 2)]]], np.float32)
    rv = orc.lidar_ray_table(5, 9, -40, 40, -10, 10)
    r, s2 = orc.raycast_lidar(rv, 10.0, "range", pos, q, tris, seg)
    hit2 = s2[0, 0] != -2
    assert hit2.any() and (~hit2).any()
    nw2, f2 = orc.raycast_lidar(rv, 10.0, "normal_world", pos, q, tris, seg)
    assert np.array_equal(nw2[0, 0][hit2], np.tile(np.float32([1, 0, 0]), (hit2.sum(), 1))) and np.all(f2[0, 0][~hit2] == -1)
    ns2, _ = orc.raycast_lidar(rv, 10.0, "normal", pos, q, tris, seg)  # world +x in the sensor frame = R(-yaw) e_x
    assert np.allclose(ns2[0, 0][hit2], [np.cos(yaw), -np.sin(yaw), 0.0], atol=1e-6) and np.all(ns2[0, 0][~hit2] == 0.0)


def test_stereo_occlusion_against_analytic_visibility(orc):
    """warp_stereo_camera_kernels.py: wall at x = 3, a small occluder plate at x = 1.5; the stereo
    partner sits `baseline` to the left of the camera (world +y).  Wall points the partner cannot
    see are -1 / -2; missed rays whose far-plane point the partner cannot see are -1 too."""
    a, b, c, d = [3, -1, -1], [3, 1, -1], [3, 1, 1], [3, -1, 1]
    e, f_, g, h = [1.5, 0.3, -0.2], [1.5, 0.6, -0.2], [1.5, 0.6, 0.2], [1.5, 0.3, 0.2]
    tris = np.array([[a + b + c, a + c + d, e + f_ + g, e + g + h]], np.float32)
    seg = np.array([[7, 7, 5, 5]], np.int32)
    W, H, hfov, far, base = 96, 72, 70.0, 10.0, 0.5
    kinv, cx, cy, pos, quat = _x_forward_camera(orc, W, H, hfov)
    depth, s = orc.raycast_stereo_camera(W, H, kinv, far, base, cx, cy, "depth", pos, quat, tris, seg)
    mono, smono = orc.raycast_camera(W, H, kinv, far, cx, cy, "depth", pos, quat, tris, seg)
    fpx = (W / 2) / np.tan(np.deg2rad(hfov / 2))
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    dy, dz = -(xs - W / 2) / fpx, -(ys - H / 2) / fpx  # ray = (1, dy, dz) t
    m = 0.04  # skip pixels within 4 cm of an edge

    def in_rect(y, z, y0, y1, z0, z1, margin):
        return (y > y0 + margin) & (y < y1 - margin) & (z > z0 + margin) & (z < z1 - margin)

    def classify(margin):
        on_occ = in_rect(1.5 * dy, 1.5 * dz, 0.3, 0.6, -0.2, 0.2, margin)
        on_wall = in_rect(3 * dy, 3 * dz, -1, 1, -1, 1, margin)
        # wall point -> partner (0, base, 0): crosses x = 1.5 half way
        wall_blocked = in_rect((3 * dy + base) / 2, 3 * dz / 2, 0.3, 0.6, -0.2, 0.2, margin)
        # far-plane point (far, far dy, far dz) -> partner: crosses the wall plane at 30 %, the occluder plane at 15 %
        far_blocked = in_rect(base + (far * dy - base) * 0.3, far * dz * 0.3, -1, 1, -1, 1, margin) | in_rect(
            base + (far * dy - base) * 0.15, far * dz * 0.15, 0.3, 0.6, -0.2, 0.2, margin)
        return on_occ, on_wall, wall_blocked, far_blocked

    occ_in, wall_in, wblk_in, fblk_in = classify(m)
    occ_out, wall_out, wblk_out, fblk_out = classify(-m)
    D, S = depth[0, 0], s[0, 0]
    sure_occ = occ_in
    sure_wall_visible = wall_in & ~occ_out & ~wblk_out
    sure_wall_hidden = wall_in & ~occ_out & wblk_in
    sure_miss_visible = ~wall_out & ~occ_out & ~fblk_out
    sure_miss_hidden = ~wall_out & ~occ_out & fblk_in
    for mask in (sure_occ, sure_wall_visible, sure_wall_hidden, sure_miss_visible, sure_miss_hidden):
        assert mask.sum() > 10
    assert np.allclose(D[sure_occ], 1.5, rtol=1e-5) and np.all(S[sure_occ] == 5)
    assert np.allclose(D[sure_wall_visible], 3.0, rtol=1e-5) and np.all(S[sure_wall_visible] == 7)
    assert np.all(D[sure_wall_hidden] == -1.0) and np.all(S[sure_wall_hidden] == -2)
    assert np.all(D[sure_miss_visible] == 1000.0) and np.all(D[sure_miss_hidden] == -1.0)
    # wherever the stereo pixel is valid it equals the monocular pixel bit for bit
    valid = D != -1.0
    assert np.array_equal(D[valid], mono[0, 0][valid]) and np.array_equal(S[valid], smono[0, 0][valid])
    # point-cloud variant: same validity pattern (range instead of depth), BVH == brute force
    pc, spc = orc.raycast_stereo_camera(W, H, kinv, far, base, cx, cy, "pointcloud_world", pos, quat, tris, seg)
    pcb, spcb = orc.raycast_stereo_camera(W, H, kinv, far, base, cx, cy, "pointcloud_world", pos, quat, tris, seg, use_bvh=True)
    assert np.array_equal(pc, pcb) and np.array_equal(spc, spcb)
    assert np.allclose(pc[0, 0][sure_wall_visible][:, 0], 3.0, rtol=1e-5) and np.all(spc[0, 0][sure_wall_hidden] == -2)


@pytest.mark.parametrize("walls", [True, False])
def test_range_and_segmentation_against_moller_trumbore_in_float64(orc, walls):
    """An independent algorithm for the same question: the oracle restates Warp's watertight Woop test in fp32,
    this check intersects every ray with every triangle by Moller-Trumbore in float64.  Ranges agree to fp32
    accuracy, the segmentation id agrees wherever the two nearest surfaces are not within that accuracy of each
    other, and so does hit / miss away from grazing rays."""
    sc = random_box_scene(2, 30, seed=9, walls=walls)  # without the room's walls many rays miss everything
    tris = orc.scene_transform(sc["tri_local"], sc["tri_asset"], sc["asset_state"])
    st = random_robot_states(2, 3, *sc["bounds"])
    pos = st[:, None, 0:3].copy()
    quat = st[:, None, 3:7].copy()
    rv = orc.lidar_ray_table(24, 64, -180, 180, -60, 60)
    far = 12.0
    rng_img, seg = orc.raycast_lidar(rv, far, "range", pos, quat, tris, sc["tri_seg"])
    tri_seg_all = np.asarray(sc["tri_seg"])

    def rotate(q, v):  # xyzw, float64
        x, y, z, w = q
        u = np.array([x, y, z])
        return v * (2 * w * w - 1) + 2 * w * np.cross(u, v) + 2 * u * (v @ u)[..., None]

    checked = 0
    for env in range(2):
        tri_seg = tri_seg_all[env] if tri_seg_all.ndim == 2 else tri_seg_all
        o = pos[env, 0].astype(np.float64)
        d = rotate(quat[env, 0].astype(np.float64), rv.reshape(-1, 3).astype(np.float64))  # [R,3] world directions
        T = tris[env].astype(np.float64).reshape(-1, 3, 3)
        a, e1, e2 = T[:, 0], T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]
        p = np.cross(d[:, None, :], e2[None])                      # [R,T,3]
        det = (p * e1[None]).sum(-1)
        ok = np.abs(det) > 1e-12
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        tv = o[None, None] - a[None]
        u = (tv * p).sum(-1) * inv
        qv = np.cross(tv, e1[None])
        v = (d[:, None, :] * qv).sum(-1) * inv
        t = (e2[None] * qv).sum(-1) * inv
        margin = np.minimum(np.minimum(u, v), 1.0 - u - v)         # barycentric distance to the triangle's border
        hit = ok & (margin >= 0) & (t > 0) & (t <= far)
        t_hit = np.where(hit, t, np.inf)
        best = t_hit.min(axis=1)
        got_r, got_s = rng_img[env, 0].reshape(-1).astype(np.float64), seg[env, 0].reshape(-1)
        # rays that clearly hit (not through an edge / grazing, not at the range limit)
        clear = np.isfinite(best) & (np.where(hit, margin, -1.0).max(axis=1) > 1e-4) & (best < far - 1e-3)
        assert clear.mean() > (0.3 if walls else 0.02)
        assert np.abs(got_r[clear] - best[clear]).max() < 2e-5 * far
        # segmentation id where the runner-up surface of a DIFFERENT asset is not within fp32 noise of the winner
        first = t_hit.argmin(axis=1)
        other = np.where(tri_seg[None, :] != tri_seg[first][:, None], t_hit, np.inf).min(axis=1)
        with np.errstate(invalid="ignore"):
            unique = clear & (other - best > 1e-4)
        assert np.array_equal(got_s[unique], tri_seg[first][unique])
        # clear misses: no triangle within a small margin of the ray
        near_miss = (ok & (margin >= -1e-4) & (t > -1e-3) & (t <= far + 1e-3)).any(axis=1)
        miss = ~near_miss
        # a miss carries the kernels' sentinel (1000, mapped to the far value by post-processing) and id -2
        assert (got_s[miss] == -2).all() and (got_r[miss] == 1000.0).all()
        assert walls or miss.mean() > 0.2
        checked += int(clear.sum()) + int(miss.sum())
    assert checked > 2000


def test_sensor_front_end_matches_the_reference_classes(orc):
    """tests/golden/sensor_frontend.npz comes from RUNNING the reference's WarpLidar / WarpCam constructors and
    WarpSensor's post-processing methods (oracle/gen_golden_sensors.py; only the Warp kernels are out of reach):
    LiDAR ray tables, camera intrinsics, noise -> range limits -> normalisation."""
    from conftest import load_golden

    g = load_golden("sensor_frontend")
    n_tables = 0
    for key in g.files:
        if key.endswith("_rays"):
            h, w, hmin, hmax, vmin, vmax = g[key.replace("_rays", "_params")]
            ours = orc.lidar_ray_table(int(h), int(w), hmin, hmax, vmin, vmax)
            assert ours.shape == g[key].shape and np.abs(ours - g[key]).max() < 1.5e-7, key  # python float64 -> fp32 store
            n_tables += 1
        if key.endswith("_K"):
            W, H, hfov, cx, cy = g[key.replace("_K", "_params")]
            kinv, ocx, ocy = orc.camera_kinv(int(W), int(H), float(hfov))
            assert (ocx, ocy) == (int(cx), int(cy)), key
            K = g[key]
            # our intrinsics are the inverse of the reference's pinhole matrix: pixel -> ray and back
            ki = np.asarray(kinv, np.float64).reshape(-1)
            Kinv = np.linalg.inv(K)
            ours = np.array([ki[0], ki[1], ki[2], ki[3]]) if ki.size == 4 else ki
            ref4 = np.array([Kinv[0, 0], Kinv[0, 2], Kinv[1, 1], Kinv[1, 2]])
            assert ours.size in (4, 16)
            got4 = ours if ours.size == 4 else np.array([ours[0], ours[2], ours[5], ours[6]])
            assert np.abs(got4 - ref4).max() < 1e-6 * (1 + np.abs(ref4).max()), key
    assert n_tables == 3
    for tag in ("depth_plain", "depth_unnormalised", "depth_noise"):
        cfg = g["pp_%s_cfg" % tag]
        px = g["pp_%s_in" % tag].copy()
        noise = cfg[7] > 0
        out = orc.sensor_postprocess(px, cfg[0], cfg[1], cfg[2], cfg[3], bool(cfg[4]),
                                     z_normal=g["pp_%s_z" % tag] if noise else None, u_dropout=g["pp_%s_u" % tag] if noise else None,
                                     std_a=cfg[8], std_b=cfg[9], std_c=cfg[10], mean_offset=cfg[11], dropout_prob=cfg[12] if noise else 0.0)
        ref = g["pp_%s_out" % tag]
        assert np.array_equal(out, ref), tag  # bit for bit, noise included (the recorded draws are the reference's)
    for tag in ("points_sensor_frame", "points_world_frame", "points_noise"):
        cfg = g["pp_%s_cfg" % tag]
        px = g["pp_%s_in" % tag].copy()
        noise = cfg[7] > 0
        world = cfg[6] > 0
        out = orc.sensor_postprocess_points(px, cfg[0], cfg[1], cfg[2], cfg[3], not world, bool(cfg[4]) and not world,
                                            z_normal=g["pp_%s_z" % tag] if noise else None, u_dropout=g["pp_%s_u" % tag] if noise else None,
                                            std_a=cfg[8], std_b=cfg[9], std_c=cfg[10], mean_offset=cfg[11],
                                            dropout_prob=cfg[12] if noise else 0.0)
        ref = g["pp_%s_out" % tag]
        assert np.array_equal(out, ref), tag  # bit for bit, noise included (the recorded draws are the reference's)


@pytest.mark.parametrize("tag", ["camera", "lidar"])
def test_sensor_mount_and_pose_match_a_real_reference_sensor(orc, tag):
    """A real WarpSensor of the reference (Warp calls inert) was initialised, reset and updated
    (oracle/gen_golden_sensors.py): its mount randomisation (rows a20: (max - min) * u + min, quat_from_euler_xyz)
    and its world pose (row a22: tf_apply and the quat_mul chain with the data-frame rotation) against the
    formulas the product's kernels and the oracle use."""
    from conftest import load_golden

    g = load_golden("sensor_frontend")
    f32 = np.float32
    cfg = g["pose_%s_cfg" % tag]
    lo_p, hi_p, lo_e, hi_e, frame_e = (cfg[0:3].astype(f32), cfg[3:6].astype(f32), np.radians(cfg[6:9]).astype(f32),
                                        np.radians(cfg[9:12]).astype(f32), np.radians(cfg[12:15]).astype(f32))
    randomize, ns = bool(cfg[15]), int(cfg[16])
    lpos, lquat = g["pose_%s_local_position" % tag], g["pose_%s_local_orientation" % tag]
    n = lpos.shape[0]
    assert lpos.shape == (n, ns, 3)
    if randomize:
        u_p, u_e = g["pose_%s_u0" % tag], g["pose_%s_u1" % tag]  # translation draws, then rotation draws
        assert np.array_equal(lpos, (hi_p - lo_p) * u_p + lo_p)
        q = orc.quat_from_euler(((hi_e - lo_e) * u_e + lo_e).astype(f32).reshape(-1, 3)).reshape(n, ns, 4)
        assert np.abs(lquat - q).max() < 2e-7
    else:
        assert not any(k.startswith("pose_%s_u" % tag) for k in g.files)
        assert np.array_equal(lpos, np.zeros_like(lpos))
        q = orc.quat_from_euler(np.tile(((lo_e + hi_e) / f32(2.0))[None], (n * ns, 1)).astype(f32)).reshape(n, ns, 4)
        assert np.abs(lquat - q).max() < 2e-7
    assert np.abs(orc.quat_from_euler(frame_e[None])[0] - g["pose_%s_frame_quat" % tag]).max() < 2e-7
    state = np.zeros((n, 13), f32)
    state[:, 0:3], state[:, 3:7] = g["pose_%s_robot_position" % tag], g["pose_%s_robot_orientation" % tag]
    pos, quat = orc.sensor_pose(state, lpos, lquat, g["pose_%s_frame_quat" % tag])
    assert np.abs(pos - g["pose_%s_sensor_position" % tag]).max() < 1e-6
    assert np.abs(quat - g["pose_%s_sensor_orientation" % tag]).max() < 3e-7
