"""Multi-link primitive obstacles from URDF files (the reference's WarpAsset for its `trees` / `thin` sets).

A URDF is reduced to a list of primitives (box / cylinder) with their pose in the asset's root-link frame
(forward kinematics over the joint tree with every joint at zero, i.e. as the reference's fixed-joint tree
assets are).  Each primitive becomes its own rigid piece of the scene (own triangles in its own frame, own
collision box), tied to the asset's pose through `agx_prims_from_assets`.  Stdlib XML only."""
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass

import numpy as np

CYLINDER_SECTIONS = 9  # 2 * 9 side + 2 * 9 cap triangles = 36 = 3 x 12: the scene works in chunks of 12 triangles


@dataclass
class Prim:
    kind: str           # "box" | "cylinder"
    dims: tuple         # box: (sx, sy, sz); cylinder: (radius, length)
    T: np.ndarray       # [4,4] primitive frame -> asset root-link frame
    link: str
    link_index: int


def rpy_matrix(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]], np.float64)


def _origin(elem):
    T = np.eye(4)
    o = elem.find("origin") if elem is not None else None
    if o is not None:
        xyz = [float(v) for v in o.get("xyz", "0 0 0").split()]
        rpy = [float(v) for v in o.get("rpy", "0 0 0").split()]
        T[:3, :3], T[:3, 3] = rpy_matrix(*rpy), xyz
    return T


def quat_xyzw_from_matrix(R):
    """rotation matrix -> unit quaternion (x, y, z, w), w >= 0"""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = [0.0, 0.0, 0.0, (R[k, j] - R[j, k]) / s]
        q[i], q[j], q[k] = 0.25 * s, (R[j, i] + R[i, j]) / s, (R[k, i] + R[i, k]) / s
    q = np.array(q, np.float64)
    q /= np.linalg.norm(q)
    return q if q[3] >= 0 else -q


def load_urdf_primitives(path, use_collision=False):
    """[Prim] of a URDF, in link order.  Geometry other than <box> / <cylinder> raises NotImplementedError."""
    root = ET.parse(path).getroot()
    links = root.findall("link")
    names = [ln.get("name") for ln in links]
    parent_of = {}
    for jn in root.findall("joint"):
        parent_of[jn.find("child").get("link")] = (jn.find("parent").get("link"), _origin(jn))
    world = {}

    def link_T(name):
        if name not in world:
            if name in parent_of:
                par, Tj = parent_of[name]
                world[name] = link_T(par) @ Tj
            else:
                world[name] = np.eye(4)
        return world[name]

    prims = []
    for idx, ln in enumerate(links):
        for el in ln.findall("collision" if use_collision else "visual"):
            geom = el.find("geometry")
            if geom is None or len(geom) == 0:
                continue
            g = geom[0]
            if g.tag == "box":
                dims = tuple(float(v) for v in g.get("size").split())
                kind = "box"
            elif g.tag == "cylinder":
                dims = (float(g.get("radius")), float(g.get("length")))
                kind = "cylinder"
            else:
                raise NotImplementedError(f"{path}: geometry <{g.tag}> of link {ln.get('name')} is not supported (box, cylinder)")
            if min(dims) <= 0.0:
                raise ValueError(f"{path}: non-positive dimensions {dims} in link {ln.get('name')}")
            prims.append(Prim(kind, dims, link_T(names[idx]) @ _origin(el), names[idx], idx))
    if not prims:
        raise ValueError(f"{path}: no box / cylinder geometry found")
    return prims


_BOX_VERTS = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]], np.float64) - 0.5
_BOX_FACES = np.array([[1, 3, 0], [4, 1, 0], [0, 3, 2], [2, 4, 0], [1, 7, 3], [5, 1, 4], [5, 7, 1], [3, 7, 2], [6, 4, 2], [2, 7, 6],
                       [6, 5, 4], [7, 5, 6]])


def tessellate(prim):
    """[T,3,3] float32 triangles in the primitive's own frame, outward normals (T = 12 box, 36 cylinder)."""
    if prim.kind == "box":
        return (_BOX_VERTS[_BOX_FACES] * np.asarray(prim.dims)).astype(np.float32)
    r, L = prim.dims
    n = CYLINDER_SECTIONS
    ang = 2 * math.pi * np.arange(n) / n
    ring = np.stack([r * np.cos(ang), r * np.sin(ang)], axis=1)
    tris = []
    top, bot = np.array([0, 0, L / 2]), np.array([0, 0, -L / 2])
    for i in range(n):
        a, b = ring[i], ring[(i + 1) % n]
        a0, b0, a1, b1 = np.r_[a, -L / 2], np.r_[b, -L / 2], np.r_[a, L / 2], np.r_[b, L / 2]
        tris += [[a0, b0, b1], [a0, b1, a1], [top, a1, b1], [bot, b0, a0]]
    return np.asarray(tris, np.float32)


def half_extents(prim):
    """half extents of the primitive's bounding box in its own frame (the collision OBB)"""
    if prim.kind == "box":
        return tuple(0.5 * d for d in prim.dims)
    r, L = prim.dims
    return (r, r, 0.5 * L)


def num_triangles(kind):
    return 12 if kind == "box" else 4 * CYLINDER_SECTIONS
