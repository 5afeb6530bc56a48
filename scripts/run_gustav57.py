"""Config-3 style run: the sfm.py driver over all 57 cameras of pose.csv on the synthetic Gustav-geometry sequence."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from datagen import gustav_scene, decompose_P
from sfm_mvs_amd import pipeline as pl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 57
K, P, feats, ids = gustav_scene(n, seed=3)
print("features per image: min %d max %d" % (min(len(f[0]) for f in feats), max(len(f[0]) for f in feats)))
t0 = time.perf_counter()
out = pl.run_sfm(feats, K)
dt = time.perf_counter() - t0
got = out["posearr"][9:].reshape(-1, 3, 4)
rerr, terr = [], []
for k in range(len(got)):
    Rg, tg = decompose_P(K, got[k]); Rw, tw = decompose_P(K, P[k])
    rerr.append(np.abs(Rg - Rw).max()); terr.append(np.linalg.norm(tg - tw) / max(1.0, np.linalg.norm(tw)))
print(f"{len(got)} cameras in {dt:.2f} s; max |dR| {max(rerr):.2e}, max rel |dt| {max(terr):.2e}; first err {out['first_error']:.4f}; "
      f"per-frame err max {max(out['errors']):.4f} median {np.median(out['errors']):.4f}; cloud {len(out['Xtot'])} pts")
print("worst frames (dR):", np.argsort(rerr)[-5:], np.sort(rerr)[-5:])
