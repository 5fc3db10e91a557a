"""Diagnostic: structural validation of the device-built BVH on the host."""
import sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scene_util import random_box_scene
from test_gpu_raycast import Scene
for n, k in ((2, 20), (3, 100)):
    sc = random_box_scene(n, k, seed=1)
    S = Scene(sc); S.build(); torch.cuda.synchronize()
    nodes = S.nodes.cpu().numpy(); NI = nodes.view(np.int32)
    nt = S.nt
    for e in range(n):
        seen = np.zeros(nt, int); visited = 0
        stack = [0]
        while stack:
            i = stack.pop(); visited += 1
            assert 0 <= i < nt - 1, ("bad node", i)
            for side, (cslot, sslot) in enumerate(((3, 11), (7, 15))):
                c, s2 = NI[e, i, cslot], NI[e, i, sslot]
                if c < 0:
                    f = ~c
                    assert 0 <= f < nt, ("bad leaf", e, i, side, c)
                    seen[f] += 1
                    if s2 >= 0:
                        assert s2 < nt, ("bad second", e, i, side, s2)
                        seen[s2] += 1
                else:
                    assert s2 == -1, ("second on internal", e, i, side, s2)
                    stack.append(c)
        print("env", e, "nt", nt, "visited nodes", visited, "of", nt - 1, "leaf coverage min/max", seen.min(), seen.max())
