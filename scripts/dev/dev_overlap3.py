"""Dev: the overlapped from-pixels job against the interpreter's GIL switch interval."""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from sfm_mvs_amd import pipeline as pl
from datagen import gustav_views
images, K, P = gustav_views(57, seed=5)
pl.run_sfm_images(images[:4], K)
for si in (0.005, 0.0005, 0.00005, 0.005, 0.0005, 0.00005):
    sys.setswitchinterval(si)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); pl.run_sfm_images(images, K); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"switchinterval {si}: " + " ".join(f"{t*1e3:.1f}" for t in ts) + f" ms  median {np.median(ts)*1e3:.1f}", flush=True)
