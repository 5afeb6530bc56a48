"""Dev: host-side profile of the 57-camera driver (where do the 0.16 s go?)."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from datagen import gustav_scene
from sfm_mvs_amd import pipeline as pl
K, P, feats, ids = gustav_scene(57, seed=3)
pl.run_sfm(feats[:4], K)
t = time.perf_counter(); pl.run_sfm(feats, K); torch.cuda.synchronize(); print("run_sfm", time.perf_counter() - t)
pr = cProfile.Profile(); pr.enable(); pl.run_sfm(feats, K); torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
