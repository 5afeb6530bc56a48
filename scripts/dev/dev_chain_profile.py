"""Dev: cProfile of the 57-camera chain from HBM-resident features (where the host time of pipeline.run_sfm goes)."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from sfm_mvs_amd import pipeline as pl
from datagen import gustav_views
images, K, P = gustav_views(57, seed=5)
feats = list(pl.FeatureStream(images, 2))
for _ in range(2): pl.run_sfm(feats, K)
torch.cuda.synchronize(); t0 = time.perf_counter(); pl.run_sfm(feats, K); torch.cuda.synchronize(); print("chain alone", (time.perf_counter() - t0) * 1e3, "ms")
pr = cProfile.Profile(); pr.enable(); pl.run_sfm(feats, K); torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
