"""TEST INFRASTRUCTURE ONLY -- a minimal emulation of `warp` (warp-lang==1.0.0, reference setup.py:18), just enough to
EXECUTE THE REFERENCE'S OWN ray-cast kernel bodies and the classes that launch them, unmodified, on the CPU:

    aerial_gym/sensors/warp/warp_kernels/warp_camera_kernels.py          DepthCameraWarpKernels.*   (5 kernels)
    aerial_gym/sensors/warp/warp_kernels/warp_lidar_kernels.py           LidarWarpKernels.*         (5 kernels)
    aerial_gym/sensors/warp/warp_kernels/warp_stereo_camera_kernels.py   StereoCameraWarpKernels.*  (4 kernels)
    aerial_gym/sensors/warp/{warp_cam, warp_lidar, warp_stereo_cam, warp_normal_faceID_cam, warp_normal_faceID_lidar}.py

warp-lang is a third-party dependency that is neither vendored under /root/reference nor installable here.  Goldens made with
this module are therefore "the reference's kernel source executed under an EMULATED Warp" -- NOT outputs of Warp itself:
`wp.inverse` (mat44, on which hip_sensor.pinhole_kinv's last-ulp behaviour rests), the Woop test and the tie-break rule are
restatements that have never been cross-checked against a real warp-lang 1.0.0 (where it can be installed, run
gen_golden_warp_kernels.py against it and diff).  What the reference's code is given instead:

  * the scalar / vector / quaternion / matrix BUILT-INS it calls, restated from Warp's published native headers
    (warp/native/vec.h, quat.h, mat.h), every operation a single IEEE binary32 operation (numpy float32 scalars; no FMA
    contraction -- Warp's CUDA build lets nvcc contract, which is compiler-version dependent and not reproducible anyway);
  * `wp.kernel`: the Python source of the decorated function is compiled as Python.  ONE source-to-source rewrite is applied:
    `wp.mesh_query_ray(mesh, start, dir, max_t, t, u, v, sign, n, f)` writes its last six arguments in Warp (C++ references);
    Python has no out-parameters, so the call is rewritten to
        __hit, t, u, v, sign, n, f = wp.mesh_query_ray(mesh, start, dir, max_t, t, u, v, sign, n, f)
    in front of the statement that contained it, and the call is replaced by `__hit` (on a miss the six values come back
    unchanged, as Warp leaves them).  Nothing else of the kernel body is touched;
  * `wp.launch(kernel, dim, inputs)`: the body is run once per thread index, `wp.tid()` returning that index;
    `capture_begin / capture_end / capture_launch` record and replay launches like the CUDA graph the reference captures;
  * `wp.from_torch / wp.to_torch`: views of the caller's torch tensors (shared memory, like Warp);
  * `wp.Mesh(points, indices, velocities)` + `wp.mesh_get`: the three arrays the reference hands over;
  * `wp.mesh_query_ray`: a BRUTE-FORCE loop over the mesh's triangles with Warp's watertight test (warp/native/intersect.h
    intersect_ray_tri_woop, restated in THIS module in numpy float32 with a correctly rounded fmaf -- `_query_brute_force`),
    0 <= t < max_t, smallest t, ties -> smallest face index.  This is the one part of the path that is a restatement of Warp,
    and the part Warp itself leaves unspecified (its answer on exact ties depends on its BVH).  It is written independently of the
    C oracle (oracle_raycast.c), which is then CHECKED against the frames produced here; `set_mesh_query(orc.mesh_query_ray)`
    swaps the C loop in (tests compare the two queries ray by ray).  The normal reported is normalize(cross(b - a, c - a)) of the
    hit face (intersect.h / mesh.h); u, v, sign are returned as 0 (no kernel of the reference reads them).

Used by oracle/gen_golden_warp_kernels.py (writes tests/golden/warp_kernels_*.npz) in the build container only.
Never imported by the product package.
"""
import ast
import inspect
import sys
import textwrap
import threading
import types

import numpy as np

f32 = np.float32
_ERR = dict(over="ignore", divide="ignore", invalid="ignore", under="ignore")


# ---------------------------------------------------------------------------------------------------------------------
# scalar types and constants
# ---------------------------------------------------------------------------------------------------------------------
float32 = np.float32
int32 = np.int32
uint64 = np.uint64


def constant(x):
    if isinstance(x, float):  # a Python float literal is a float32 in Warp
        return f32(x)
    return x


# ---------------------------------------------------------------------------------------------------------------------
# vec3 / quat / mat44  (warp/native/vec.h, quat.h, mat.h)
# ---------------------------------------------------------------------------------------------------------------------
class vec3:
    __slots__ = ("c",)
    _n = 3

    def __init__(self, *a):
        if len(a) == 0:
            self.c = (f32(0.0), f32(0.0), f32(0.0))
        elif len(a) == 1 and isinstance(a[0], vec3):
            self.c = a[0].c
        elif len(a) == 3:
            self.c = (f32(a[0]), f32(a[1]), f32(a[2]))
        else:
            raise TypeError("vec3() takes 0 or 3 scalars")

    def __getitem__(self, i):
        return self.c[i]

    def __add__(self, o):
        with np.errstate(**_ERR):
            return vec3(self.c[0] + o.c[0], self.c[1] + o.c[1], self.c[2] + o.c[2])

    def __sub__(self, o):
        with np.errstate(**_ERR):
            return vec3(self.c[0] - o.c[0], self.c[1] - o.c[1], self.c[2] - o.c[2])

    def __mul__(self, s):  # vec.h mul(vec, scalar): component * s
        s = f32(s)
        with np.errstate(**_ERR):
            return vec3(self.c[0] * s, self.c[1] * s, self.c[2] * s)

    def __rmul__(self, s):  # vec.h mul(scalar, vec) = mul(vec, scalar)
        return self.__mul__(s)

    def __truediv__(self, s):  # vec.h div(vec, scalar): component / s
        s = f32(s)
        with np.errstate(**_ERR):
            return vec3(self.c[0] / s, self.c[1] / s, self.c[2] / s)

    def __neg__(self):
        return vec3(-self.c[0], -self.c[1], -self.c[2])

    def __repr__(self):
        return "vec3(%r, %r, %r)" % tuple(float(x) for x in self.c)


class quat:
    """xyzw (quat.h quat_t: x, y, z, w)"""
    __slots__ = ("x", "y", "z", "w")
    _n = 4

    def __init__(self, *a):
        if len(a) == 0:
            a = (0.0, 0.0, 0.0, 0.0)
        self.x, self.y, self.z, self.w = (f32(v) for v in a)

    def __getitem__(self, i):
        return (self.x, self.y, self.z, self.w)[i]


class mat44:
    """row-major 4 x 4 float32 (mat.h mat_t<4,4,float>; the constructor takes the 16 entries row by row)"""
    _n = 16

    def __init__(self, *a):
        if len(a) == 1:
            a = np.asarray(a[0]).reshape(-1)
        if len(a) != 16:
            raise TypeError("mat44() takes 16 scalars")
        self.m = np.array([f32(v) for v in a], dtype=np.float32).reshape(4, 4)

    def __getitem__(self, ij):
        return self.m[ij]


def dot(a, b):
    """vec.h dot(): a0 b0 + a1 b1 + a2 b2, summed in index order"""
    with np.errstate(**_ERR):
        r = a.c[0] * b.c[0]
        r = r + a.c[1] * b.c[1]
        r = r + a.c[2] * b.c[2]
    return r


def length(a):
    """vec.h length(): sqrt(dot(a, a))"""
    return np.sqrt(dot(a, a))


def normalize(a):
    """vec.h normalize(): l = length(a); l > kEps (0) ? a / l : 0"""
    if isinstance(a, quat):
        raise NotImplementedError("normalize(quat) is not used by the reference's kernels")
    l = length(a)
    if l > f32(0.0):
        return a / l
    return vec3()


def cross(a, b):
    """vec.h cross()"""
    with np.errstate(**_ERR):
        return vec3(a.c[1] * b.c[2] - a.c[2] * b.c[1], a.c[2] * b.c[0] - a.c[0] * b.c[2], a.c[0] * b.c[1] - a.c[1] * b.c[0])


def quat_rotate(q, x):
    """quat.h quat_rotate(q, x):
         c = 2 w w - 1;  d = 2 (q.xyz . x)
         x_i c + q_i d + (q.xyz X x)_i w 2"""
    two, one = f32(2.0), f32(1.0)
    with np.errstate(**_ERR):
        c = two * q.w * q.w - one
        d = two * (q.x * x.c[0] + q.y * x.c[1] + q.z * x.c[2])
        return vec3(
            x.c[0] * c + q.x * d + (q.y * x.c[2] - q.z * x.c[1]) * q.w * two,
            x.c[1] * c + q.y * d + (q.z * x.c[0] - q.x * x.c[2]) * q.w * two,
            x.c[2] * c + q.z * d + (q.x * x.c[1] - q.y * x.c[0]) * q.w * two,
        )


def quat_inverse(q):
    """quat.h quat_inverse(): (-x, -y, -z, w)"""
    return quat(-q.x, -q.y, -q.z, q.w)


def transform_vector(m, v):
    """mat.h transform_vector(mat44, vec3) = (m * (v, 0)).xyz with mul(mat, vec): r = col_0 v_0; r += col_i v_i (i = 1..3)"""
    b = (v.c[0], v.c[1], v.c[2], f32(0.0))
    out = []
    with np.errstate(**_ERR):
        for r in range(3):
            acc = m.m[r, 0] * b[0]
            for i in range(1, 4):
                acc = acc + m.m[r, i] * b[i]
            out.append(acc)
    return vec3(*out)


def inverse(m):
    """mat.h inverse(mat44), "adapted from USD GfMatrix4f::Inverse()": cofactor expansion.  USD's float version keeps the
    matrix entries and the cofactors of the first two columns in the matrix's scalar type (float32) and the 2x2 determinants,
    the cofactors of the last two columns, the determinant and its reciprocal in double; every cofactor is multiplied by the
    reciprocal in double and rounded to float32 once.  (Restated third-party code: the one place of this emulation where
    the precision of an intermediate is taken from memory of the published source rather than from its operation order.
    For the pinhole matrices the reference builds -- warp_cam.py:43-60 -- the double and the all-float32 reading can differ
    in the last bit of K_inv[0][0] / [1][1] / [0][2] / [1][2]; the entries travel to the kernels as data, so the ray-cast
    fixtures do not depend on which one the product's host code picks, only `test_kinv_matches` does.)"""
    if not isinstance(m, mat44):
        raise TypeError("inverse() emulates the mat44 overload only")
    f64 = np.float64
    with np.errstate(**_ERR):
        x = m.m  # float32 entries; float32 * float32 rounds to float32 before it is widened, as in C++
        x00, x01, x10, x11, x20, x21, x30, x31 = x[0, 0], x[0, 1], x[1, 0], x[1, 1], x[2, 0], x[2, 1], x[3, 0], x[3, 1]
        y01 = f64(x00 * x11 - x10 * x01)
        y02 = f64(x00 * x21 - x20 * x01)
        y03 = f64(x00 * x31 - x30 * x01)
        y12 = f64(x10 * x21 - x20 * x11)
        y13 = f64(x10 * x31 - x30 * x11)
        y23 = f64(x20 * x31 - x30 * x21)
        x02, x03, x12, x13, x22, x23, x32, x33 = x[0, 2], x[0, 3], x[1, 2], x[1, 3], x[2, 2], x[2, 3], x[3, 2], x[3, 3]
        z33 = f64(x02) * y12 - f64(x12) * y02 + f64(x22) * y01  # float * double -> double
        z23 = f64(x12) * y03 - f64(x32) * y01 - f64(x02) * y13
        z13 = f64(x02) * y23 - f64(x22) * y03 + f64(x32) * y02
        z03 = f64(x22) * y13 - f64(x32) * y12 - f64(x12) * y23
        z32 = f64(x13) * y02 - f64(x23) * y01 - f64(x03) * y12
        z22 = f64(x03) * y13 - f64(x13) * y03 + f64(x33) * y01
        z12 = f64(x23) * y03 - f64(x33) * y02 - f64(x03) * y23
        z02 = f64(x13) * y23 - f64(x23) * y13 + f64(x33) * y12
        y01 = f64(x02 * x13 - x12 * x03)
        y02 = f64(x02 * x23 - x22 * x03)
        y03 = f64(x02 * x33 - x32 * x03)
        y12 = f64(x12 * x23 - x22 * x13)
        y13 = f64(x12 * x33 - x32 * x13)
        y23 = f64(x22 * x33 - x32 * x23)
        z30 = f32(f64(x11) * y02 - f64(x21) * y01 - f64(x01) * y12)  # float cofactors: computed in double, stored as float
        z20 = f32(f64(x01) * y13 - f64(x11) * y03 + f64(x31) * y01)
        z10 = f32(f64(x21) * y03 - f64(x31) * y02 - f64(x01) * y23)
        z00 = f32(f64(x11) * y23 - f64(x21) * y13 + f64(x31) * y12)
        z31 = f32(f64(x00) * y12 - f64(x10) * y02 + f64(x20) * y01)
        z21 = f32(f64(x10) * y03 - f64(x30) * y01 - f64(x00) * y13)
        z11 = f32(f64(x00) * y23 - f64(x20) * y03 + f64(x30) * y02)
        z01 = f32(f64(x20) * y13 - f64(x30) * y12 - f64(x10) * y23)
        det = f64(x30 * z30) + f64(x31 * z31) + f64(x32) * z32 + f64(x33) * z33
        if not abs(det) > 0.0:
            return mat44(*([0.0] * 16))
        rcp = f64(1.0) / det
        e = lambda z: f32(f64(z) * rcp)  # noqa: E731
        return mat44(e(z00), e(z10), e(z20), e(z30),
                     e(z01), e(z11), e(z21), e(z31),
                     e(z02), e(z12), e(z22), e(z32),
                     e(z03), e(z13), e(z23), e(z33))


# ---------------------------------------------------------------------------------------------------------------------
# arrays (views of torch tensors)
# ---------------------------------------------------------------------------------------------------------------------
class _ArrayAnnotation:
    def __init__(self, dtype=None, ndim=1):
        self.dtype, self.ndim = dtype, ndim


class array:
    """As a call inside an annotation, `wp.array(dtype=..., ndim=...)` describes a kernel argument; as an object it wraps a
    numpy view (from_torch).  Element type vec3 / quat / mat44 folds the trailing axis."""

    def __new__(cls, *a, dtype=None, ndim=1, **k):
        if not a and not k.get("_data"):
            return _ArrayAnnotation(dtype, ndim)
        return super().__new__(cls)

    def __init__(self, data=None, dtype=None, torch_tensor=None, **k):
        if not isinstance(data, np.ndarray):  # wp.array(list_of_mesh_ids, dtype=wp.uint64)
            data = np.asarray(data, dtype=np.float32 if dtype is float else dtype)
        self.np = data
        self.dtype = dtype
        self.torch_tensor = torch_tensor
        self._vec = getattr(dtype, "_n", None)
        self.shape = data.shape[:-1] if self._vec else data.shape
        self.ndim = len(self.shape)

    def _idx(self, i):
        i = i if isinstance(i, tuple) else (i,)
        if len(i) != self.ndim:
            raise IndexError(f"{len(i)} indices into a {self.ndim}-d warp array")
        for v, n in zip(i, self.shape):
            if not 0 <= int(v) < n:
                raise IndexError(f"index {i} out of range for shape {self.shape}")
        return tuple(int(v) for v in i)

    def __getitem__(self, i):
        v = self.np[self._idx(i)]
        if self._vec:
            return self.dtype(*v)
        return v  # numpy scalar of the array's dtype

    def __setitem__(self, i, value):
        i = self._idx(i)
        if self._vec:
            if not isinstance(value, self.dtype):
                raise TypeError(f"storing {type(value).__name__} into an array of {self.dtype.__name__}")
            self.np[i] = value.c if isinstance(value, vec3) else (value.x, value.y, value.z, value.w)
        else:
            self.np[i] = self.np.dtype.type(value)


def array2d(dtype=None):
    return _ArrayAnnotation(dtype, 2)


_NP_OF = {np.float32: np.float32, np.int32: np.int32, np.uint64: np.uint64}


def from_torch(t, dtype=None):
    if t is None:
        return None
    a = t.detach().numpy()  # shares memory with the tensor
    if dtype in (vec3, quat):
        if a.shape[-1] != dtype._n or a.dtype != np.float32:
            raise TypeError(f"from_torch(dtype={dtype.__name__}): trailing axis {a.shape[-1]}, dtype {a.dtype}")
    elif dtype is not None and a.dtype != np.dtype(dtype):
        raise TypeError(f"from_torch: tensor is {a.dtype}, asked for {np.dtype(dtype)}")
    return array(a, dtype=dtype if dtype in (vec3, quat) else (a.dtype.type), torch_tensor=t, _data=True)


def to_torch(a):
    return a.torch_tensor


# ---------------------------------------------------------------------------------------------------------------------
# meshes
# ---------------------------------------------------------------------------------------------------------------------
_MESHES = {}
_TRI_CACHE = {}


def _fma32(a, b, c):
    """fmaf(a, b, c) on float32 arrays, correctly rounded: the product is exact in float64 (24 + 24 bits), the sum is rounded to ODD
    in float64 (TwoSum gives the exact error) and then once to float32 -- round-to-odd with 53 >= 24 + 2 bits makes the double
    rounding innocuous."""
    r = a.astype(np.float64) * b.astype(np.float64)
    c = c.astype(np.float64)
    with np.errstate(**_ERR):
        s = r + c
        bb = s - r
        e = (r - (s - bb)) + (c - bb)
    fix = (e != 0.0) & np.isfinite(s) & ((s.view(np.int64) & 1) == 0)
    if fix.any():
        si = s.view(np.int64).copy()
        away = (e > 0.0) == (s > 0.0)  # the exact sum lies further from zero than s: next representable magnitude up, else down
        si[fix & away] += 1
        si[fix & ~away] -= 1
        s = si.view(np.float64)
    return s.astype(np.float32)


def _diff_product(a, b, c, d):
    """intersect.h diff_product(): a b - c d with FMA error compensation (Kahan)"""
    with np.errstate(**_ERR):
        cd = c * d
        diff = _fma32(a, b, -cd)
        error = _fma32(-c, d, cd)
        return diff + error


def _query_brute_force(o, d, max_t, tris):
    """wp.mesh_query_ray's closest hit by a loop over ALL triangles of the mesh (vectorised over the triangles): Warp's watertight
    test, warp/native/intersect.h intersect_ray_tri_woop -- dominant axis kz = first strict maximum of |dir|, kx / ky the next two
    (swapped when dir[kz] < 0), shear S = (dir[kx], dir[ky], 1) / dir[kz], vertices translated by the origin and sheared, the
    three edge functions by diff_product with a DOUBLE-precision fallback when one is exactly 0, rejection on mixed signs, on a zero
    determinant and on sign(T) != sign(det), t = T * (1 / det) -- accepting 0 <= t < max_t, keeping the smallest t and on an exact
    tie the smallest face index (mesh.h keeps whichever its BVH visits first; see the module docstring).
    Written here from Warp's published source, independently of oracle/oracle_raycast.c: the fixtures this module produces are what
    the C oracle is CHECKED against (tests/test_oracle_warp_kernels.py)."""
    d = [f32(x) for x in d]
    o = [f32(x) for x in o]
    ax, ay, az = abs(d[0]), abs(d[1]), abs(d[2])
    kz = (0 if ax > az else 2) if ax > ay else (1 if ay > az else 2)
    kx = 0 if kz == 2 else kz + 1
    ky = 0 if kx == 2 else kx + 1
    if d[kz] < f32(0.0):
        kx, ky = ky, kx
    with np.errstate(**_ERR):
        Sx, Sy, Sz = d[kx] / d[kz], d[ky] / d[kz], f32(1.0) / d[kz]
        A = tris[:, 0:3] - np.array(o, np.float32)
        B = tris[:, 3:6] - np.array(o, np.float32)
        Cc = tris[:, 6:9] - np.array(o, np.float32)
        Ax, Ay = A[:, kx] - Sx * A[:, kz], A[:, ky] - Sy * A[:, kz]
        Bx, By = B[:, kx] - Sx * B[:, kz], B[:, ky] - Sy * B[:, kz]
        Cx, Cy = Cc[:, kx] - Sx * Cc[:, kz], Cc[:, ky] - Sy * Cc[:, kz]
        U = _diff_product(Cx, By, Cy, Bx)
        V = _diff_product(Ax, Cy, Ay, Cx)
        W = _diff_product(Bx, Ay, By, Ax)
        z = (U == 0.0) | (V == 0.0) | (W == 0.0)
        if z.any():
            f64 = np.float64
            U = np.where(z, (Cx.astype(f64) * By.astype(f64) - Cy.astype(f64) * Bx.astype(f64)).astype(np.float32), U)
            V = np.where(z, (Ax.astype(f64) * Cy.astype(f64) - Ay.astype(f64) * Cx.astype(f64)).astype(np.float32), V)
            W = np.where(z, (Bx.astype(f64) * Ay.astype(f64) - By.astype(f64) * Ax.astype(f64)).astype(np.float32), W)
        mixed = ((U < 0.0) | (V < 0.0) | (W < 0.0)) & ((U > 0.0) | (V > 0.0) | (W > 0.0))
        det = U + V + W
        Az, Bz, Cz = Sz * A[:, kz], Sz * B[:, kz], Sz * Cc[:, kz]
        T = U * Az + V * Bz + W * Cz
        # xorf(T, sign(det)) < 0  <=>  T != 0 and T, det of opposite sign (a zero of either sign passes)
        behind = (np.signbit(T) != np.signbit(det)) & (T != 0.0)
        t = T * (f32(1.0) / det)
        ok = ~mixed & (det != 0.0) & ~behind & (t >= 0.0) & (t < f32(max_t))
    if not ok.any():
        return False, f32(0.0), -1
    tt = np.where(ok, t, np.float32(np.inf))
    face = int(np.argmin(tt))  # the first minimum: smallest face index on an exact tie
    return True, f32(tt[face]), face


_QUERY = [_query_brute_force]  # (o, d, max_t, tris [T, 9] float32) -> (hit, t, face)


def set_mesh_query(fn):
    """replace the closest-hit query (e.g. by the C oracle's brute force, to time the generator or to cross-check this module)"""
    _QUERY[0] = fn


class Mesh:
    def __init__(self, points=None, indices=None, velocities=None, **k):
        if points is None or indices is None:
            raise TypeError("Mesh(points, indices[, velocities])")
        self.points, self.indices, self.velocities = points, indices, velocities
        self.id = np.uint64(0x1000 + len(_MESHES))
        _MESHES[int(self.id)] = self

    def refit(self):
        """Warp rebuilds the bounds of its BVH from the (aliased) points array; the brute-force query reads the points
        themselves, so only the gathered-triangle cache has to go."""
        _TRI_CACHE.pop(int(self.id), None)

    def triangles(self):
        t = _TRI_CACHE.get(int(self.id))
        if t is None:
            idx = np.asarray(self.indices.np, dtype=np.int64).reshape(-1, 3)
            t = np.ascontiguousarray(self.points.np[idx].reshape(-1, 9), dtype=np.float32)
            _TRI_CACHE[int(self.id)] = t
        return t


def mesh_get(mesh_id):
    return _MESHES[int(mesh_id)]


def mesh_query_ray(mesh_id, start, direction, max_t, t, u, v, sign, n, f):
    """-> (hit, t, u, v, sign, n, f): the out-parameters of Warp's builtin as return values (see the module docstring)."""
    mesh = _MESHES[int(mesh_id)]
    tris = mesh.triangles()
    hit, th, face = _QUERY[0](start.c, direction.c, f32(max_t), tris)
    if not hit:
        return False, t, u, v, sign, n, f
    a, b, c = vec3(*tris[face, 0:3]), vec3(*tris[face, 3:6]), vec3(*tris[face, 6:9])
    return True, f32(th), f32(0.0), f32(0.0), f32(0.0), normalize(cross(b - a, c - a)), int(face)


# ---------------------------------------------------------------------------------------------------------------------
# kernels and launches
# ---------------------------------------------------------------------------------------------------------------------
_TLS = threading.local()


def tid():
    t = _TLS.tid
    return t[0] if len(t) == 1 else t


class _HoistMeshQuery(ast.NodeTransformer):
    """wp.mesh_query_ray(mesh, o, d, max_t, t, u, v, sign, n, f)  ->  tuple assignment in front of the statement + `__hitK`"""

    def __init__(self):
        self.k = 0
        self.pending = []

    @staticmethod
    def _is_query(node):
        return (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "mesh_query_ray"
                and isinstance(node.func.value, ast.Name) and node.func.value.id == "wp")

    def visit_Call(self, node):
        self.generic_visit(node)
        if not self._is_query(node):
            return node
        if len(node.args) != 10 or not all(isinstance(a, ast.Name) for a in node.args[4:]):
            raise NotImplementedError("mesh_query_ray: the six out-parameters must be plain variables")
        name = "__hit%d" % self.k
        self.k += 1
        targets = [ast.Name(id=name, ctx=ast.Store())] + [ast.Name(id=a.id, ctx=ast.Store()) for a in node.args[4:]]
        self.pending.append(ast.Assign(targets=[ast.Tuple(elts=targets, ctx=ast.Store())], value=node))
        return ast.Name(id=name, ctx=ast.Load())

    def _block(self, stmts):
        out = []
        for s in stmts:
            if isinstance(s, ast.If):
                s.test = self.visit(s.test)  # queries inside the condition are evaluated just before the `if`
                out.extend(self.pending)
                self.pending = []
                s.body, s.orelse = self._block(s.body), self._block(s.orelse)
                out.append(s)
            elif isinstance(s, (ast.For, ast.While, ast.With, ast.Try)):
                raise NotImplementedError("control flow other than `if` around mesh_query_ray is not emulated")
            else:
                s = self.visit(s)
                out.extend(self.pending)
                self.pending = []
                if isinstance(s, ast.Expr) and isinstance(s.value, ast.Name) and s.value.id.startswith("__hit"):
                    continue  # a bare `wp.mesh_query_ray(...)` statement: the assignment is all of it
                out.append(s)
        return out

    def rewrite(self, fn_def):
        fn_def.body = self._block(fn_def.body)
        return fn_def


class Kernel:
    def __init__(self, fn):
        self.py = fn
        self.name = fn.__qualname__
        self.sig = inspect.signature(fn)
        src = textwrap.dedent(inspect.getsource(fn))
        tree = ast.parse(src)
        fd = tree.body[0]
        fd.decorator_list = []
        self.rewrites = 0
        h = _HoistMeshQuery()
        h.rewrite(fd)
        self.rewrites = h.k
        for a in fd.args.args:  # the annotations were evaluated when the reference module was imported
            a.annotation = None
        ast.fix_missing_locations(tree)
        ns = {}
        exec(compile(tree, inspect.getsourcefile(fn) or "<kernel>", "exec"), fn.__globals__, ns)
        self.fn = ns[fd.name]
        self.source_lines = inspect.getsourcelines(fn)[1]

    def convert(self, inputs):
        params = list(self.sig.parameters.values())
        if len(inputs) != len(params):
            raise TypeError(f"{self.name}: {len(inputs)} inputs for {len(params)} parameters")
        out = []
        for p, v in zip(params, inputs):
            an = p.annotation
            if isinstance(an, _ArrayAnnotation):
                if isinstance(v, (list, tuple)):  # a list of mesh ids (the reference hands CONST_WARP_MESH_ID_LIST through wp.array)
                    v = array(np.asarray(v, dtype=an.dtype), dtype=an.dtype, _data=True)
                if not isinstance(v, array):
                    raise TypeError(f"{self.name}: parameter {p.name} wants a warp array, got {type(v).__name__}")
                if v.ndim != an.ndim:
                    raise TypeError(f"{self.name}: parameter {p.name} wants ndim {an.ndim}, got {v.ndim}")
                if an.dtype in (vec3, quat) and v.dtype is not an.dtype:
                    raise TypeError(f"{self.name}: parameter {p.name} wants {an.dtype.__name__} elements")
                out.append(v)
            elif an is float:
                out.append(f32(v))
            elif an is int:
                out.append(int(v))
            elif an is bool:
                out.append(bool(v))
            elif an is mat44:
                if not isinstance(v, mat44):
                    raise TypeError(f"{self.name}: parameter {p.name} wants a mat44")
                out.append(v)
            else:
                raise TypeError(f"{self.name}: annotation {an!r} of {p.name} is not emulated")
        return out


def kernel(fn):
    return Kernel(fn)


UNDEFINED_READS = []  # (kernel name, tid, variable): threads that read a variable their control path never assigned
LAUNCH_LOG = []       # (kernel name, dim): every launch that ran
_CAPTURE = [None]


def _run(k, dim, args):
    LAUNCH_LOG.append((k.name, tuple(dim)))
    _TRI_CACHE.clear()  # the points may have been rewritten between launches
    for t in np.ndindex(*dim):
        _TLS.tid = t
        try:
            k.fn(*args)
        except UnboundLocalError as e:  # Warp's generated C++ declares every variable up front: the read is of an uninitialised value
            UNDEFINED_READS.append((k.name, t, str(e)))


def launch(kernel=None, dim=None, inputs=(), outputs=(), device=None, **kw):  # noqa: A002
    if not isinstance(kernel, Kernel):
        raise TypeError("launch: not a wp.kernel")
    dim = (dim,) if isinstance(dim, int) else tuple(int(d) for d in dim)
    args = kernel.convert(list(inputs) + list(outputs))
    if _CAPTURE[0] is not None:
        _CAPTURE[0].append((kernel, dim, args))  # stream capture: recorded, not run
        return
    _run(kernel, dim, args)


class Graph:
    def __init__(self, launches):
        self.launches = launches


def capture_begin(device=None, **kw):
    if _CAPTURE[0] is not None:
        raise RuntimeError("capture_begin inside a capture")
    _CAPTURE[0] = []


def capture_end(device=None, **kw):
    g, _CAPTURE[0] = Graph(_CAPTURE[0]), None
    return g


def capture_launch(graph, **kw):
    for k, dim, args in graph.launches:
        _run(k, dim, args)


def init():
    pass


def install():
    """register this module as `warp` (before the reference's sensor modules are imported)"""
    me = sys.modules[__name__]
    if sys.modules.get("warp") not in (None, me):
        raise RuntimeError("another `warp` module is already imported")
    sys.modules["warp"] = me
    return me


def __getattr__(name):  # PEP 562: nothing is silently inert
    raise AttributeError(f"wp.{name} is not emulated (oracle/wp_emul.py)")
