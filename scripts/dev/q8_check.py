"""Dev: the integer body on 8-bit QUANTISED float data (SFM_KNN_Q8=1) against the oracle on a few shapes / data kinds,
then the batch-of-8 step time beside the 16-bit body's (one box)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sfm_mvs_amd import ops
from oracle import oracle as O
O.lib()


def check(name, q, t, filt="auto"):
    gi, gd, st = ops.knn2(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda(), return_stats=True, filter=filt)
    torch.cuda.synchronize()
    gi, gd, st = gi.cpu().numpy(), gd.cpu().numpy(), st.cpu().numpy()
    wi, wd = O.knn2(q, t, nthreads=64)
    bad_i = int((gi != wi).any(1).sum())
    bad_d = int((gd.view(np.uint32) != wd.view(np.uint32)).any(1).sum())
    print(f"{name:28s} {q.shape[0]:6d} x {t.shape[0]:6d}  idx rows differ {bad_i}  dist rows differ {bad_d}  stats {st.tolist()}", flush=True)
    return bad_i + bad_d


rng = np.random.default_rng(0)
bad = 0
for nq, nt in [(1, 2), (5, 3), (64, 64), (129, 1000), (777, 1234), (3000, 2500), (10000, 10000), (100, 20000), (20000, 96)]:
    bad += check("uniform", rng.random((nq, 128), dtype=np.float32), rng.random((nt, 128), dtype=np.float32))
bad += check("normal*37.5", (rng.standard_normal((500, 128)) * 37.5).astype(np.float32), (rng.standard_normal((900, 128)) * 37.5 + 3).astype(np.float32))
t = rng.random((4000, 128), dtype=np.float32)
q = t[rng.integers(0, 4000, 3000)] + (rng.standard_normal((3000, 128)) * 1e-3).astype(np.float32)
bad += check("near-duplicates", q, t)
t = np.repeat(rng.random((50, 128), dtype=np.float32), 40, axis=0)
bad += check("duplicated trains", rng.random((300, 128), dtype=np.float32), t)
q = rng.random((2000, 128), dtype=np.float32); q[::7, 5] = 40.0
t = rng.random((3000, 128), dtype=np.float32); t[::11, 9] = -25.0
bad += check("outliers", q, t)
bad += check("tiny", rng.random((800, 128), dtype=np.float32) * 1e-6, rng.random((900, 128), dtype=np.float32) * 1e-6)
bad += check("sift-like u8", rng.integers(0, 120, (2000, 128)).astype(np.float32), rng.integers(0, 120, (3000, 128)).astype(np.float32))
print("TOTAL MISMATCHING ROWS", bad, flush=True)
