import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from sfm_mvs_amd import ops
from oracle import oracle as O
rng = np.random.default_rng(7)
q = np.zeros((64, 128), np.float32)
t = np.full((900, 128), 200.0, np.float32)
flip = rng.integers(0, 128, 900)
which = rng.integers(0, 3, 900)
for r in range(900):
    if which[r] == 2:
        t[r, flip[r]] = 199.0
t[::7, 5] = 201.0
q[1::2, 3] = 1.0
for nt in (900, 896, 100, 32, 33):
    tt = t[:nt].copy()
    gi, gd, st = [o.cpu().numpy() for o in ops.knn2(torch.from_numpy(q).cuda(), torch.from_numpy(tt).cuda(), return_stats=True)]
    wi, wd = O.knn2(q, tt, nthreads=4)
    print(nt, "stats", st, "gpu", gi[0], gd[0] ** 2, "want", wi[0], wd[0] ** 2, "equal", np.array_equal(gi, wi))
print("---- repeat / diff detail")
for nt in (896, 900):
    tt = t[:nt].copy()
    for rep in range(3):
        gi, gd, st = [o.cpu().numpy() for o in ops.knn2(torch.from_numpy(q).cuda(), torch.from_numpy(tt).cuda(), return_stats=True)]
        wi, wd = O.knn2(q, tt, nthreads=4)
        bad = np.where((gi != wi).any(1))[0]
        print(nt, rep, "ndiff", len(bad), "rows", bad[:8], "gpu", gi[bad[:3]].tolist(), (gd[bad[:3]] ** 2).tolist(), "want", wi[bad[:3]].tolist(), (wd[bad[:3]] ** 2).tolist())
from datagen import sift_like
rng = np.random.default_rng(129 * 31 + 1000)
qq, tt = sift_like(rng, 129), sift_like(rng, 1000)
for rep in range(3):
    gi, gd, st = [o.cpu().numpy() for o in ops.knn2(torch.from_numpy(qq).cuda(), torch.from_numpy(tt).cuda(), return_stats=True)]
    wi, wd = O.knn2(qq, tt, nthreads=4)
    bad = np.where((gi != wi).any(1))[0]
    print("sift129x1000", rep, st, "ndiff", len(bad), "rows", bad[:8], "gpu", gi[bad[:3]].tolist(), (gd[bad[:3]] ** 2).tolist(), "want", wi[bad[:3]].tolist(), (wd[bad[:3]] ** 2).tolist())
