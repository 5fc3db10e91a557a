import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import numpy as np, torch
from scene_util import random_box_scene
from test_gpu_raycast import Scene
sc = random_box_scene(4, 50, seed=2)
S = Scene(sc)
S.build(); n0 = S.nodes.clone()
for rep in range(3):
    S.build()
    d = (S.nodes != n0)
    print("rep", rep, "mismatch count", int(d.sum()), "of", d.numel())
    if d.any():
        idx = d.nonzero()[:10]
        for e, nd, c in idx.tolist():
            print(e, nd, c, float(n0[e, nd, c]), float(S.nodes[e, nd, c]), n0[e, nd, c].view(torch.int32).item(), S.nodes[e, nd, c].view(torch.int32).item())
        nd_set = sorted(set((e, nd) for e, nd, c in d.nonzero().tolist()))
        print("nodes affected", len(nd_set), nd_set[:10])
