"""Dev: where the from-pixels job's time goes when features are produced ahead of the driver (pipeline.FeatureStream)."""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from sfm_mvs_amd import pipeline as pl
from datagen import gustav_views
images, K, P = gustav_views(57, seed=5)
pl.run_sfm_images(images[:4], K)
def t(f, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best, r
def only_features():
    fs = pl.FeatureStream(images, 2)
    out = list(fs); fs.close(); return out
ta, feats = t(only_features)
tb, _ = t(lambda: pl.run_sfm(feats, K))
tc, _ = t(lambda: pl.run_sfm_images(images, K))
pinned = [torch.from_numpy(im).pin_memory() for im in images]
td, _ = t(lambda: pl.run_sfm_images(pinned, K))
te, _ = t(lambda: list(pl.FeatureStream(pinned, 2)))
print(f"features alone (FeatureStream drained) {ta*1e3:.1f} ms | chain alone from HBM features {tb*1e3:.1f} ms | overlapped {tc*1e3:.1f} ms | overlapped, pinned frames {td*1e3:.1f} ms | features alone, pinned {te*1e3:.1f} ms")
