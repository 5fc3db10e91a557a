"""Writes tests/fixtures/assets/trees13/tree_13.urdf: a SYNTHETIC 13-link tree (a trunk + 12 branches on fixed joints, every link
one <cylinder>), numbers from a seeded generator -- the SHAPE of the reference's `trees` set (13 single-cylinder links, star of
fixed joints on the first link; resources/models/environment_assets/trees/tree_*.urdf) without any of its content, so that the
GPU box, which has no reference tree, exercises forest_env at the reference's size: 13 x 128 cylinder triangles + 35 objects + a
floor = 2148 triangles per env."""
import os

import numpy as np

rng = np.random.default_rng(20260926)
out = ['<?xml version="1.0"?>', '<robot name="tree_13">']
joints = []
for i in range(13):
    if i == 0:
        r, L = 0.16, 6.5
        xyz, rpy = (0.0, 0.0, L / 2), (0.0, 0.05, 0.3)
    else:
        r, L = float(rng.uniform(0.03, 0.09)), float(rng.uniform(0.8, 2.6))
        xyz, rpy = (0.0, 0.0, L / 2), (0.0, 0.0, 0.0)
        a = float(rng.uniform(0, 2 * np.pi))
        joints.append((i, (0.1 * np.cos(a), 0.1 * np.sin(a), float(rng.uniform(1.5, 6.0))), (float(rng.uniform(0.4, 1.3)), float(rng.uniform(-0.4, 0.4)), a)))
    out.append(f'  <link name="limb_{i}">')
    for tag in ("visual", "collision"):
        out.append(f'    <{tag}><origin xyz="{xyz[0]:.6f} {xyz[1]:.6f} {xyz[2]:.6f}" rpy="{rpy[0]:.6f} {rpy[1]:.6f} {rpy[2]:.6f}"/>'
                   f'<geometry><cylinder radius="{r:.6f}" length="{L:.6f}"/></geometry></{tag}>')
    out.append("  </link>")
for i, xyz, rpy in joints:
    out.append(f'  <joint name="j_{i}" type="fixed"><parent link="limb_0"/><child link="limb_{i}"/>'
               f'<origin xyz="{xyz[0]:.6f} {xyz[1]:.6f} {xyz[2]:.6f}" rpy="{rpy[0]:.6f} {rpy[1]:.6f} {rpy[2]:.6f}"/></joint>')
out.append("</robot>")
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "trees13", "tree_13.urdf")
open(path, "w").write("\n".join(out) + "\n")
print(path)
