"""Where the HOST time of the 57-camera chain goes: cProfile of pipeline.run_sfm on the synthetic Gustav sequence of bench.py's
`sfm` leg (features resident).  The chain is host-bound: ~0.7 ms per camera for ~0.1 ms of device work.
  python scripts/profile_chain.py [top N]"""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from datagen import gustav_scene
from sfm_mvs_amd import pipeline as pl
top = int(sys.argv[1]) if len(sys.argv) > 1 else 45
K, P, feats, ids = gustav_scene(57, seed=3)
pl.run_sfm(feats[:4], K)
for _ in range(2):
    t0 = time.perf_counter(); pl.run_sfm(feats, K); torch.cuda.synchronize(); print(f"run_sfm: {(time.perf_counter() - t0) * 1e3:.1f} ms")
pr = cProfile.Profile()
pr.enable(); pl.run_sfm(feats, K); torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(top)
