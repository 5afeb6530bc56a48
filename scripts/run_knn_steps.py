"""Profiling driver: N steps of the config-2 hot path (10k x 10k KNN + ratio) and nothing else."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_mvs_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
q = torch.rand((nq, 128), generator=torch.Generator().manual_seed(0)).cuda()
t = torch.rand((nt, 128), generator=torch.Generator().manual_seed(1)).cuda()
pm = ops.PairMatcher(nq, nt, q.device)
for _ in range(n):
    pm.run(q, t)
torch.cuda.synchronize()
print("done", pm.stats.cpu().tolist())
