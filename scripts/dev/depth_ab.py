"""Dev: the batched KNN step (8 distinct 10k x 10k pairs per launch set, two image sets rotating) at pipeline depths 1..4, uniform
float data (quantised integer body) and SIFT-like u8 data (exact integer body), alternating so that all arms share the box's state."""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from sfm_mvs_amd import ops
from datagen import planted_pair
dev = torch.device("cuda")
nq = nt = 10000
uni = [[(torch.rand((nq, 128), generator=torch.Generator().manual_seed(2 * (8 * s + b))).to(dev),
         torch.rand((nt, 128), generator=torch.Generator().manual_seed(2 * (8 * s + b) + 1)).to(dev)) for b in range(8)] for s in range(2)]
rng = np.random.default_rng(0)
sift = [[tuple(torch.from_numpy(a).to(dev) for a in planted_pair(rng, nq, nt, 0.3)[:2]) for b in range(8)] for s in range(2)]
pipes = {d: ops.BatchPipeline(nq, nt, dev, ratio=0.70, depth=d, batch=8) for d in (1, 2, 3, 4)}
def run(pipe, sets, n):
    for i in range(n):
        for q, t in sets[i % 2]:
            pipe.submit(q, t, after=False)
    pipe.flush(); pipe.synchronize()
for d, p in pipes.items():
    run(p, uni, 3); run(p, sift, 3)
run(pipes[3], uni, 300)                      # clock ramp
for rep in range(4):
    for name, sets in (("uniform", uni), ("sift-like", sift)):
        line = f"rep {rep} {name:9s}:"
        for d, p in pipes.items():
            run(p, sets, 20)
            t0 = time.perf_counter(); run(p, sets, 100); dt = (time.perf_counter() - t0) / 100
            line += f"  depth {d}: {dt * 1e3:.4f} ms ({8 * nq * nt / dt:.3e}/s)"
        print(line, flush=True)
