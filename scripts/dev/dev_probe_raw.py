"""Dev: raw timings of ops.streams_overlap in a fresh process (chain = high priority, candidates normal)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from sfm_mvs_amd import ops
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
buf = torch.zeros(1 << 26, device=dev); small = buf[:64]
buf[64:].mul_(1.0); small.mul_(1.0); torch.cuda.synchronize()
def raw(a, b):
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        e0.record(a); buf[64:].mul_(1.0); buf[64:].mul_(1.0); ea.record(a)
    with torch.cuda.stream(b):
        small.mul_(1.0); eb.record(b)
    torch.cuda.synchronize()
    return e0.elapsed_time(ea), e0.elapsed_time(eb)
chain = torch.cuda.Stream(device=dev, priority=-1)
cands = [torch.cuda.Stream(device=dev) for _ in range(10)]
for k, c in enumerate(cands):
    for rep in range(2):
        la, lb = raw(chain, c); ra, rb = raw(c, chain)
        print(f"candidate {k} rep {rep}: long on chain {la*1e3:.0f} us, short on cand done at {lb*1e3:.0f} us | long on cand {ra*1e3:.0f} us, short on chain done at {rb*1e3:.0f} us", flush=True)
