"""Dev: the integer body on 8-bit QUANTISED float data (SFM_KNN_Q8=1) against the oracle on a few shapes / data kinds,
then the batch-of-8 step time beside the 16-bit body's (one box)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sfm_mvs_amd import ops
from oracle import oracle as O
O.lib()


def check(name, q, t, filt="auto"):
    gi, gd, st = ops.knn2(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda(), return_stats=True, filter=filt)
    torch.cuda.synchronize()
    gi, gd, st = gi.cpu().numpy(), gd.cpu().numpy(), st.cpu().numpy()
    wi, wd = O.knn2(q, t, nthreads=64)
    bad_i = int((gi != wi).any(1).sum())
    bad_d = int((gd.view(np.uint32) != wd.view(np.uint32)).any(1).sum())
    print(f"{name:28s} {q.shape[0]:6d} x {t.shape[0]:6d}  idx rows differ {bad_i}  dist rows differ {bad_d}  stats {st.tolist()}", flush=True)
    return bad_i + bad_d


rng = np.random.default_rng(0)
bad = 0
for nq, nt in [(1, 2), (5, 3), (64, 64), (129, 1000), (777, 1234), (3000, 2500), (10000, 10000), (100, 20000), (20000, 96)]:
    bad += check("uniform", rng.random((nq, 128), dtype=np.float32), rng.random((nt, 128), dtype=np.float32))
bad += check("normal*37.5", (rng.standard_normal((500, 128)) * 37.5).astype(np.float32), (rng.standard_normal((900, 128)) * 37.5 + 3).astype(np.float32))
t = rng.random((4000, 128), dtype=np.float32)
q = t[rng.integers(0, 4000, 3000)] + (rng.standard_normal((3000, 128)) * 1e-3).astype(np.float32)
bad += check("near-duplicates", q, t)
t = np.repeat(rng.random((50, 128), dtype=np.float32), 40, axis=0)
bad += check("duplicated trains", rng.random((300, 128), dtype=np.float32), t)
q = rng.random((2000, 128), dtype=np.float32); q[::7, 5] = 40.0
t = rng.random((3000, 128), dtype=np.float32); t[::11, 9] = -25.0
bad += check("outliers", q, t)
bad += check("tiny", rng.random((800, 128), dtype=np.float32) * 1e-6, rng.random((900, 128), dtype=np.float32) * 1e-6)
bad += check("sift-like u8", rng.integers(0, 120, (2000, 128)).astype(np.float32), rng.integers(0, 120, (3000, 128)).astype(np.float32))
print("TOTAL MISMATCHING ROWS", bad, flush=True)

# ---- fallbacks, repairs, mixed batches (through the batched entry point)
def check_batch(name, pairs, filt="auto"):
    nq, nt = pairs[0][0].shape[0], pairs[0][1].shape[0]
    bm = ops.BatchMatcher(nq, nt, "cuda", ratio=0.70, batch=len(pairs), filter=filt)
    dev = [(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()) for q, t in pairs]
    bm.run(dev)
    torch.cuda.synchronize()
    tot = 0
    for b, (q, t) in enumerate(pairs):
        wi, wd = O.knn2(q, t, nthreads=64)
        gi, gd = bm.idx[b].cpu().numpy(), bm.dist[b].cpu().numpy()
        tot += int((gi != wi).any(1).sum()) + int((gd.view(np.uint32) != wd.view(np.uint32)).any(1).sum())
    print(f"{name:40s} batch {len(pairs)} {nq} x {nt}: mismatching rows {tot}  stats {bm.stats[:len(pairs)].cpu().numpy().tolist()}", flush=True)
    return tot


U = lambda n: rng.random((n, 128), dtype=np.float32)
G = lambda n: rng.standard_normal((n, 128)).astype(np.float32)
S = lambda n: rng.integers(0, 120, (n, 128)).astype(np.float32)
nq, nt = 1500, 2100
bad2 = 0
bad2 += check_batch("uniform x4", [(U(nq), U(nt)) for _ in range(4)])
bad2 += check_batch("uniform + gaussian (repair)", [(U(nq), U(nt)), (G(nq), G(nt)), (U(nq), U(nt))])
bad2 += check_batch("uniform + u8 (one grid each)", [(U(nq), U(nt)), (S(nq), S(nt)), (U(nq) * 7 - 3, U(nt) * 7 - 3)])
t = U(nt); t[5:900:7, 3] = 9.0                                   # outliers the sample does not see (rows 0, nt/16, ...)
bad2 += check_batch("hidden outliers (repair)", [(U(nq), t), (U(nq), U(nt))])
t = U(nt); t[1] = 0.5; t[2] = 0.0; t[3] = 1.0                     # constant rows at mid-range and at the ends: init product's range
bad2 += check_batch("constant rows", [(U(nq), t)])
bad2 += check_batch("gaussian x2 (no attempt)", [(G(nq), G(nt)), (G(nq) * 5, G(nt) * 5)])
bad2 += check_batch("noquant variant", [(U(nq), U(nt))], filt="noquant")
bad2 += check_batch("half variant", [(U(nq), U(nt))], filt="half")
q = U(nq); q[7, 11] = np.inf; t = U(nt); t[9, 3] = np.nan
print("non-finite values (no parity claim, must not hang):", end=" ")
try:
    check_batch("non-finite", [(q, t)])
except Exception as e:
    print("raised", e)
print("TOTAL MISMATCHING ROWS (batches)", bad2, flush=True)
