"""Dev: batched KNN step timing: B pairs per launch set, S streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_mvs_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1
nq = nt = 10000
q = torch.rand((nq, 128), generator=torch.Generator().manual_seed(0)).cuda()
t = torch.rand((nt, 128), generator=torch.Generator().manual_seed(1)).cuda()
bms = [ops.BatchMatcher(nq, nt, q.device, batch=B) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
pairs = [(q, t)] * B
for i in range(2 * S):
    with torch.cuda.stream(streams[i % S]):
        bms[i % S].run(pairs)
torch.cuda.synchronize()
ops.profile_read(0); ops.profile_read(1)
if S == 1:
    ops.profile_enable(True)
t0 = time.perf_counter()
for i in range(n):
    with torch.cuda.stream(streams[i % S]):
        bms[i % S].run(pairs)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (n * B)
f_ms, f_n = ops.profile_read(0)
r_ms, r_n = ops.profile_read(1)
ops.profile_enable(False)
print(f"batch {B} streams {S}: {dt*1e3:.4f} ms per pair  {nq*nt/dt:.3e} dist/s  filter {f_ms/max(f_n,1):.4f} ms per launch = {f_ms/max(f_n,1)/B:.4f} per pair, refine {r_ms/max(r_n,1)/B:.4f} per pair; stats {bms[0].stats[0].tolist()}")
