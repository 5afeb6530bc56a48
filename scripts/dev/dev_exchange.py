"""Dev: where the single-rank exchange path of bench.py loses time: CPU enqueue time per step vs GPU time per step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sfm_mvs_amd import ops, sharded
nq = nt = 10000; B = 8; depth = 3
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "exchange"
if len(sys.argv) > 2:          # also initialise a one-rank process group like bench.py does
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group(sys.argv[2], device_id=dev if sys.argv[2] == "nccl" else None, world_size=1, rank=0)
    if len(sys.argv) > 3:
        x = torch.zeros(4, device=dev); dist.all_reduce(x); torch.cuda.synchronize()
q = torch.rand((nq, 128), generator=torch.Generator().manual_seed(0)).to(dev)
t = torch.rand((nt, 128), generator=torch.Generator().manual_seed(1)).to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "exchange"
pipe = ops.BatchPipeline(nq, nt, dev, ratio=0.70, depth=depth, batch=B)
ex = sharded.BatchedExchange((2, nq, 2), torch.int32, dev, batch=8, nbuf=depth + 1) if mode != "plain" else None
def step():
    if ex is None:
        for _ in range(B): pipe.submit(q, t, after=False)
        return
    for _ in range(B):
        slot, ev = ex.next_slot()
        pipe.submit(q, t, after=ev if (ev is not None and mode != "noevent") else False, result=slot)
        if ex.commit():
            pipe.flush()
            if mode != "nogather": ex.flush(pipe.streams)
            else: ex.cur, ex.fill = (ex.cur + 1) % len(ex.local), 0
for _ in range(300): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
cpu = 0.0
for _ in range(200):
    c0 = time.perf_counter(); step(); cpu += time.perf_counter() - c0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{mode}: {dt/200*1e6:.1f} us per step wall ({dt/1600*1e6:.2f} per pair), CPU enqueue {cpu/200*1e6:.1f} us per step")
