"""CPU: known-answer tests for the two oracle functions no reference source can pin (`parity unpinned` in
oracle/oracle_dynamics.c): the rigid-body integrator that stands in for PhysX and the sphere/box collision
flag.  They pin the SCHEME the header names -- semi-implicit Euler with the force applied before gravity,
multiplicative damping, velocity clamps, exponential-map orientation update, gyroscopic term in the body
frame -- against closed forms evaluated in float64, and the predicate against hand-computed distances."""
import numpy as np
import pytest
from conftest import golden_params, load_golden, rel_err


@pytest.fixture()
def quad(orc):
    pd = dict(golden_params(load_golden("step_quad_position")))
    pd.update(linear_damping=0.0, angular_damping=0.0, max_linear_velocity=1e6, max_angular_velocity=1e6)
    return pd


def _state(n=1, p=(0, 0, 0), q=(0, 0, 0, 1), v=(0, 0, 0), w=(0, 0, 0)):
    s = np.zeros((n, 13), np.float32)
    s[:, 0:3], s[:, 3:7], s[:, 7:10], s[:, 10:13] = p, q, v, w
    return s


def test_translation_is_semi_implicit_euler(orc, quad):
    P = orc.make_params(quad)
    dt, m, g = float(P.dt), float(P.mass), np.array(list(P.gravity), np.float64)
    F = np.array([0.3, -0.2, 1.7 * m * 9.81], np.float64)  # body == world (identity orientation)
    p0, v0 = np.array([1.0, -2.0, 3.0]), np.array([0.5, 0.25, -1.0])
    s = _state(p=p0, v=v0)
    wrench = np.concatenate([F, np.zeros(3)])[None].astype(np.float32)
    n = 200
    for _ in range(n):
        orc.integrate(P, s, wrench)
    a = F / m + g
    v_n = v0 + n * a * dt
    p_n = p0 + n * v0 * dt + a * dt * dt * n * (n + 1) / 2  # positions use the NEW velocity of each step
    assert rel_err(s[0, 7:10], v_n) < 2e-6
    assert rel_err(s[0, 0:3], p_n) < 2e-6
    explicit = p0 + n * v0 * dt + a * dt * dt * n * (n - 1) / 2  # what an explicit Euler step would give
    assert np.abs(s[0, 0:3] - explicit).max() > 1e-3
    assert rel_err(s[0, 3:7], [0, 0, 0, 1]) < 1e-7  # no torque, no spin: orientation untouched


def test_body_frame_force_is_rotated_into_the_world(orc, quad):
    P = orc.make_params(quad)
    dt, m, g = float(P.dt), float(P.mass), np.array(list(P.gravity), np.float64)
    q = np.array([0.0, np.sin(np.pi / 4), 0.0, np.cos(np.pi / 4)])  # +90 deg about y: body z -> world x
    s = _state(q=q)
    orc.integrate(P, s, np.array([[0, 0, 2.0, 0, 0, 0]], np.float32))
    assert rel_err(s[0, 7:10], np.array([2.0 / m, 0, 0]) * dt + g * dt) < 1e-6


def test_spin_about_a_principal_axis_and_constant_torque(orc, quad):
    quad = dict(quad, inertia=[0.01, 0, 0, 0, 0.02, 0, 0, 0, 0.03])
    quad.pop("inertia_inv", None)
    P = orc.make_params(quad)
    dt = float(P.dt)
    w0, tau, n = 3.0, 0.006, 300
    s = _state(w=(0, 0, w0))
    wrench = np.array([[0, 0, 0, 0, 0, tau]], np.float32)
    for _ in range(n):
        orc.integrate(P, s, wrench)
    alpha = tau / 0.03
    w_n = w0 + n * alpha * dt
    theta = n * w0 * dt + alpha * dt * dt * n * (n + 1) / 2  # the exponential map uses the new angular velocity
    # the angular velocity makes a world -> body -> world round trip through the fp32 quaternion every step
    # (|q| = 1 to 1e-7 only): 1e-7 per step systematically, 300 steps
    assert rel_err(s[0, 10:13], [0, 0, w_n]) < 1e-4
    assert rel_err(s[0, 3:7], [0, 0, np.sin(theta / 2), np.cos(theta / 2)]) < 2e-4
    assert abs(np.linalg.norm(s[0, 3:7].astype(np.float64)) - 1.0) < 1e-6


def test_gyroscopic_term_keeps_the_angular_momentum(orc, quad):
    """Torque-free tumbling of an asymmetric body: the world-frame angular momentum L = R J R^T w is conserved by
    the continuous equations.  The first-order scheme drifts by about a percent over half a second; dropping the
    w x Jw term (body rates constant) would move L by 45 % over the same run."""
    J = np.diag([0.01, 0.02, 0.035])
    quad = dict(quad, inertia=J.reshape(-1).tolist())
    quad.pop("inertia_inv", None)
    P = orc.make_params(quad)
    dt, n = float(P.dt), 50
    wb0 = np.array([2.0, 5.0, 1.0])
    s = _state(w=wb0)

    def rot(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    L0 = J @ wb0  # identity orientation
    zero = np.zeros((1, 6), np.float32)
    for _ in range(n):
        orc.integrate(P, s, zero)
    R = rot(s[0, 3:7].astype(np.float64))
    L1 = R @ J @ R.T @ s[0, 10:13].astype(np.float64)
    assert np.abs(R.T @ s[0, 10:13] - wb0).max() > 0.3  # the body rates did change: it tumbles
    drift = np.linalg.norm(L1 - L0) / np.linalg.norm(L0)
    assert drift < 0.02
    # the same run with constant body rates (no gyroscopic term), closed form: rotation about wb0 by |wb0| n dt
    th, ax = np.linalg.norm(wb0) * n * dt, wb0 / np.linalg.norm(wb0)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R2 = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    assert np.linalg.norm(R2 @ J @ wb0 - L0) / np.linalg.norm(L0) > 20 * drift


def test_damping_and_velocity_clamps(orc, quad):
    quad = dict(quad, linear_damping=0.4, angular_damping=0.7, gravity=[0.0, 0.0, 0.0])
    P = orc.make_params(quad)
    dt, n = float(P.dt), 150
    s = _state(v=(1.0, -2.0, 0.5), w=(0.0, 0.0, 4.0))
    zero = np.zeros((1, 6), np.float32)
    for _ in range(n):
        orc.integrate(P, s, zero)
    assert rel_err(s[0, 7:10], np.array([1.0, -2.0, 0.5]) * (1 - 0.4 * dt) ** n) < 5e-6
    assert rel_err(s[0, 10:13], np.array([0.0, 0.0, 4.0]) * (1 - 0.7 * dt) ** n) < 1e-4  # fp32 quaternion round trip, see above
    quad = dict(quad, linear_damping=0.0, angular_damping=0.0, max_linear_velocity=3.0, max_angular_velocity=2.0)
    P = orc.make_params(quad)
    s = _state(v=(30.0, 40.0, 0.0), w=(0.0, 0.0, 10.0))  # spin about a principal axis: no gyroscopic torque
    orc.integrate(P, s, zero)
    assert rel_err(s[0, 7:10], [1.8, 2.4, 0.0]) < 1e-6 and rel_err(s[0, 10:13], [0.0, 0.0, 2.0]) < 1e-6
    assert rel_err(s[0, 0:3], np.array([1.8, 2.4, 0.0]) * dt) < 1e-6  # the pose moves with the clamped velocity


def test_sphere_box_collision_flag(orc):
    r = 0.25
    half = np.array([0.5, 1.0, 0.2])

    def hit(p, q=(0, 0, 0, 1), centre=(0, 0, 0)):
        boxes = np.zeros((1, 1, 10), np.float32)
        boxes[0, 0, 0:3], boxes[0, 0, 3:7], boxes[0, 0, 7:10] = centre, q, half
        crashes = np.zeros(1, np.uint8)
        orc.collide_sphere_boxes(r, _state(p=p), boxes, crashes)
        return bool(crashes[0])

    eps = 1e-4
    assert hit((0.5 + r - eps, 0, 0)) and not hit((0.5 + r + eps, 0, 0))            # face
    assert hit((0, 0, -(0.2 + r - eps))) and not hit((0, 0, -(0.2 + r + eps)))
    a = r / np.sqrt(2.0)                                                              # edge: distance a * sqrt(2)
    assert hit((0.5 + a - eps, 1.0 + a - eps, 0)) and not hit((0.5 + a + eps, 1.0 + a + eps, 0))
    a = r / np.sqrt(3.0)                                                              # corner
    assert hit((0.5 + a - eps, 1.0 + a - eps, 0.2 + a - eps)) and not hit((0.5 + a + eps, 1.0 + a + eps, 0.2 + a + eps))
    assert hit((0.1, -0.3, 0.05))                                                     # centre inside the box
    # the same box turned 90 deg about z and moved: its long side now lies along x
    qz = (0.0, 0.0, np.sin(np.pi / 4), np.cos(np.pi / 4))
    assert hit((3.0 + 1.0 + r - eps, 2.0, 0), qz, (3, 2, 0)) and not hit((3.0 + 1.0 + r + eps, 2.0, 0), qz, (3, 2, 0))
    assert not hit((3.0, 2.0 + 0.5 + r + eps, 0), qz, (3, 2, 0)) and hit((3.0, 2.0 + 0.5 + r - eps, 0), qz, (3, 2, 0))
    # the flag is sticky (accumulated over the sub-steps of an env step) and any box of the env counts
    boxes = np.zeros((1, 2, 10), np.float32)
    boxes[0, :, 3:7] = (0, 0, 0, 1)
    boxes[0, 0, 0:3], boxes[0, 0, 7:10] = (10, 0, 0), half
    boxes[0, 1, 0:3], boxes[0, 1, 7:10] = (0, 0, 0), half
    crashes = np.zeros(1, np.uint8)
    orc.collide_sphere_boxes(r, _state(p=(0.6, 0, 0)), boxes, crashes)
    assert crashes[0] == 1
    orc.collide_sphere_boxes(r, _state(p=(5, 5, 5)), boxes, crashes)
    assert crashes[0] == 1


def test_matrix_to_quaternion_against_an_independent_implementation(orc):
    """pytorch3d's matrix_to_quaternion (a dependency the reference does not pin or vendor) as restated in the
    oracle and in oracle/ref_shells.py, against scipy's Rotation.from_matrix: same rotation up to the sign the
    controller does not care about, on all four branches of the algorithm (largest of w, x, y, z)."""
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(11)
    rots = [Rotation.random(2000, random_state=5)]
    for axis in np.eye(3):  # half turns and near-half turns about each axis: the x / y / z branches
        ang = np.pi - np.abs(rng.normal(0, 1e-3, 200))
        rots.append(Rotation.from_rotvec(axis[None] * ang[:, None]))
    rots.append(Rotation.from_rotvec(rng.normal(0, 1e-4, (200, 3))))  # near identity: the w branch at its edge
    R = np.concatenate([r.as_matrix() for r in rots]).astype(np.float32)
    q = orc.rotmat_to_quat(R).astype(np.float64)
    q_ref = Rotation.from_matrix(R.astype(np.float64)).as_quat()  # xyzw
    assert np.abs(np.linalg.norm(q, axis=1) - 1.0).max() < 5e-6
    dots = np.abs((q * q_ref).sum(axis=1))
    assert dots.min() > 1.0 - 1e-6
    branches = np.argmax(np.abs(q[:, [3, 0, 1, 2]]), axis=1)
    assert set(branches.tolist()) == {0, 1, 2, 3}
    import torch
    from ref_shells import _matrix_to_quaternion

    q_shell = _matrix_to_quaternion(torch.from_numpy(R).reshape(-1, 3, 3)).numpy()[:, [1, 2, 3, 0]]  # wxyz -> xyzw
    assert np.abs(q_shell - q).max() < 2e-6  # the shell the goldens were generated through is the same function


def test_kinematic_obstacles_follow_their_twist(orc):
    """f4: obstacle twists -> poses over k sub-steps (orc_assets_integrate; Isaac Gym moves these bodies inside PhysX,
    nothing to pin against).  Constant twist has a closed form: p + k dt v, and a rotation about w by k dt |w|."""
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(3)
    n, K, dt, k = 4, 5, 0.01, 10
    st = np.zeros((n, K, 13), np.float32)
    st[..., 0:3] = rng.normal(0, 3, (n, K, 3))
    q0 = Rotation.random(n * K, random_state=1)
    st[..., 3:7] = q0.as_quat().reshape(n, K, 4)
    tw = rng.normal(0, 1.0, (n, K, 6)).astype(np.float32)
    tw[0, 0, 3:6] = 0.0  # no spin at all
    tw[0, 1, :] = 0.0    # a static obstacle
    before = st.copy()
    orc.assets_integrate(st, tw, dt, k)
    # k fp32 additions at |p| up to ~8 m: a few ulps of 8
    assert np.abs(st[..., 0:3] - (before[..., 0:3].astype(np.float64) + k * dt * tw[..., 0:3])).max() < 6e-6
    w = tw[..., 3:6].reshape(-1, 3).astype(np.float64)
    want = (Rotation.from_rotvec(w * k * dt) * q0).as_quat()  # world-frame angular velocity: left multiplication
    got = st[..., 3:7].reshape(-1, 4).astype(np.float64)
    assert (np.abs((got * want).sum(axis=1)) > 1 - 1e-6).all()
    assert np.array_equal(st[..., 7:13], tw)                      # the twist is what the state then reports as velocity
    assert np.array_equal(st[0, 1, 0:7], before[0, 1, 0:7])       # zero twist: pose untouched, bit for bit
