import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from sfm_mvs_amd import ops
nq = nt = 10000; B = 4; S = 3
q = torch.rand((nq, 128), generator=torch.Generator().manual_seed(0)).cuda()
t = torch.rand((nt, 128), generator=torch.Generator().manual_seed(1)).cuda()
bms = [ops.BatchMatcher(nq, nt, q.device, batch=B) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
pairs = [(q, t)] * B
for i in range(2 * S):
    with torch.cuda.stream(streams[i % S]): bms[i % S].run(pairs)
torch.cuda.synchronize()
# chunks of 25 sets each, synchronised between (like bench's region) and not
for rep in range(3):
    out = []
    for chunk in range(8):
        t0 = time.perf_counter()
        for i in range(25):
            with torch.cuda.stream(streams[i % S]): bms[i % S].run(pairs)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 100 * 1e6)
    print("25-set chunks, us per pair:", " ".join(f"{x:.1f}" for x in out))
    time.sleep(0.5)
t0 = time.perf_counter()
for i in range(200):
    with torch.cuda.stream(streams[i % S]): bms[i % S].run(pairs)
torch.cuda.synchronize()
print("200 sets: %.1f us per pair" % ((time.perf_counter() - t0) / 800 * 1e6))
