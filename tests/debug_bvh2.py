import sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scene_util import random_box_scene
from test_gpu_raycast import Scene
for n, k, walls in ((1, 1, False), (1, 2, False), (1, 4, False), (1, 20, False), (1, 20, True)):
    sc = random_box_scene(n, k, seed=1, walls=walls)
    S = Scene(sc); print("building", n, k, walls, "nt", S.nt, flush=True)
    S.build(); torch.cuda.synchronize()
    NI = S.nodes.cpu().numpy().view(np.int32)
    print(NI[0, :, [3, 7, 11, 15]].T[:12].tolist(), flush=True)
