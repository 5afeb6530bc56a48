"""Dev: is a two-deep KNN pipeline slow when its two streams share a hardware queue?  Fresh pipelines, stream handles printed."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("HWQ", "8"))
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from sfm_mvs_amd import ops
dev = torch.device("cuda")
nq = nt = 10000
uni = [[(torch.rand((nq, 128), generator=torch.Generator().manual_seed(2 * (8 * s + b))).to(dev),
         torch.rand((nt, 128), generator=torch.Generator().manual_seed(2 * (8 * s + b) + 1)).to(dev)) for b in range(8)] for s in range(2)]
def run(pipe, n):
    for i in range(n):
        for q, t in uni[i % 2]:
            pipe.submit(q, t, after=False)
    pipe.flush(); pipe.synchronize()
warm = ops.BatchPipeline(nq, nt, dev, depth=3, batch=8); run(warm, 300)
keep = []
DEPTH = int(os.environ.get("DEPTH", "2"))
for trial in range(int(os.environ.get("TRIALS", "16"))):
    p = ops.BatchPipeline(nq, nt, dev, depth=DEPTH, batch=8)
    keep.append(p); keep[:] = keep[-2:]              # (workspaces are 1.5 GB per launch set: keep only the last two pipelines alive)
    run(p, 20)
    t0 = time.perf_counter(); run(p, 100); dt = (time.perf_counter() - t0) / 100
    print(f"trial {trial:2d}: {dt*1e3:.4f} ms   streams {[hex(s.cuda_stream) for s in p.streams]}", flush=True)
    if trial % 3 == 2: torch.cuda.Stream()            # skip one pool stream now and then: changes the pairing
