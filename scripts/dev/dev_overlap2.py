"""Dev: the overlapped from-pixels job at FeatureStream depths 1..3 and lookaheads."""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from sfm_mvs_amd import pipeline as pl
from datagen import gustav_views
images, K, P = gustav_views(57, seed=5)
pl.run_sfm_images(images[:4], K)
orig = pl.FeatureStream.__init__
for depth, look in ((3, 8), (2, 8), (1, 8), (1, 3), (2, 4), (3, 8), (1, 8)):
    def init(self, images, downscale=2, depth_=depth, lookahead_=look):
        orig(self, images, downscale, depth_, lookahead_)
    pl.FeatureStream.__init__ = init
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); pl.run_sfm_images(images, K); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"depth {depth} lookahead {look}: " + " ".join(f"{t*1e3:.1f}" for t in ts) + " ms", flush=True)
