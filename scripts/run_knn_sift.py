"""Profiling driver: the config-2 shape on SIFT-like integer descriptors (SURVEY 8d distribution (ii): 30 % planted twins) —
B DISTINCT pairs per launch set, two sets alternating, S launch sets in flight; prints the pipelined rate.
  python scripts/run_knn_sift.py [steps] [B] [S] [nq] [nt] [filter]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sfm_mvs_amd import ops
from datagen import planted_pair
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
nt = int(sys.argv[5]) if len(sys.argv) > 5 else 10000
filt = sys.argv[6] if len(sys.argv) > 6 else "auto"
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
sets = []
def one_pair():
    if os.environ.get("SFM_RANDINT"):                        # dev: uniform integers 0..119 instead of SIFT-like rows
        return (torch.from_numpy(rng.integers(0, 120, (nq, 128)).astype(np.float32)).to(dev), torch.from_numpy(rng.integers(0, 120, (nt, 128)).astype(np.float32)).to(dev))
    return tuple(torch.from_numpy(a).to(dev) for a in planted_pair(rng, nq, nt, 0.3)[:2])
for s in range(2):
    sets.append([one_pair()] * B if os.environ.get("SFM_SAME") else [one_pair() for _ in range(B)])
bms = [ops.BatchMatcher(nq, nt, dev, batch=B, filter=filt) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
for i in range(int(os.environ.get("SFM_WARM", "60")) * S):   # load + clock ramp
    with torch.cuda.stream(streams[i % S]):
        bms[i % S].run(sets[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    with torch.cuda.stream(streams[i % S]):
        bms[i % S].run(sets[i % 2])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (n * B)
print(f"done sift-like {nq}x{nt} filter={filt}: batch {B} x {S} in flight: {dt*1e3:.4f} ms per pair  {nq*nt/dt:.3e} dist/s  stats {bms[0].stats[0].tolist()}")
