"""Randomised parity sweep: sfm_match_l2_f32 (through PairMatcher) against the CPU oracle on random shapes and data
families, including the degenerate ones that force rescans.  Usage: python scripts/fuzz_knn.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sfm_mvs_amd import ops
from oracle import oracle as O
from datagen import planted_pair, sift_like

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = len(sys.argv) > 3 and sys.argv[3] == "big"       # also long train sets (many substreams / candidate records per query)
rng = np.random.default_rng(seed)


def make(kind, nq, nt):
    if kind == "uniform":
        return rng.random((nq, 128), dtype=np.float32), rng.random((nt, 128), dtype=np.float32)
    if kind == "normal_scaled":
        s = np.float32(10.0 ** rng.uniform(-4, 4))
        return (rng.standard_normal((nq, 128)) * s).astype(np.float32), (rng.standard_normal((nt, 128)) * s + s).astype(np.float32)
    if kind == "sift":
        return sift_like(rng, nq), sift_like(rng, nt)
    if kind == "planted":
        q, t, _ = planted_pair(rng, nq, max(nt, 2), 0.3)
        return q, t
    if kind == "duplicates":
        base = rng.random((max(1, nt // 7), 128), dtype=np.float32)
        t = np.tile(base, (8, 1))[:nt]
        q = base[rng.integers(0, len(base), nq)] + np.float32(1e-3) * rng.standard_normal((nq, 128)).astype(np.float32)
        return q.astype(np.float32), t
    if kind == "near_ties":
        base = rng.random((1, 128), dtype=np.float32)
        t = (base * (1 + np.float32(1e-6) * rng.standard_normal((nt, 1)).astype(np.float32))).astype(np.float32)
        return rng.random((nq, 128), dtype=np.float32), t
    if kind == "unit":
        q = rng.standard_normal((nq, 128)); t = rng.standard_normal((nt, 128))
        return (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32), (t / np.linalg.norm(t, axis=1, keepdims=True)).astype(np.float32)
    if kind == "mixed_magnitude":
        q = rng.random((nq, 128), dtype=np.float32); t = rng.random((nt, 128), dtype=np.float32)
        t[:: max(1, nt // 9)] *= np.float32(1e-4); q[::3] *= np.float32(100.0)
        return q, t
    raise ValueError(kind)


kinds = ["uniform", "normal_scaled", "sift", "planted", "duplicates", "near_ties", "unit", "mixed_magnitude"]
t_end = time.time() + budget
cases = fails = batched = 0
modes = {}
while time.time() < t_end:
    kind = kinds[cases % len(kinds)]
    nq = int(rng.choice([1, 3, 17, 64, 255, 257, 1000, 2049, 5000]) if rng.random() < 0.5 else rng.integers(1, 6000))
    nt = int(rng.choice([1, 2, 31, 33, 512, 1023, 4097, 9000]) if rng.random() < 0.5 else rng.integers(1, 12000))
    if big and cases % 3 == 0:
        nq, nt = int(rng.integers(1, 700)), int(rng.integers(20000, 70000))
    if kind in ("duplicates", "near_ties"):
        nq, nt = min(nq, 600), min(nt, 3000)             # every stream is rescanned: keep the exact work bounded
    q, t = make(kind, nq, nt)
    nq, nt = len(q), len(t)
    variant = ["auto", "auto", "auto", "split", "f32", "auto", "lds"][cases % 7]
    if variant != "f32" and cases % 4 == 1:
        # a batch of 2..8 pairs of this shape in ONE launch set (sfm_match_batch_l2_f32), data families mixed: the batch runs
        # the most general arithmetic mode any pair needs, every pair must still equal the oracle
        B = int(rng.integers(2, 9))
        nqb, ntb = min(nq, 1500), min(nt, 4000)
        pairs = [make(kinds[(cases + 3 * b) % len(kinds)] if b else kind, nqb, ntb) for b in range(B)]
        pairs = [(q[:nqb], t[:ntb]) for q, t in pairs if len(q) >= nqb and len(t) >= ntb]
        nqb, ntb = min(len(p[0]) for p in pairs), min(len(p[1]) for p in pairs)
        pairs = [(np.ascontiguousarray(q[:nqb]), np.ascontiguousarray(t[:ntb])) for q, t in pairs]
        bm = ops.BatchMatcher(nqb, ntb, "cuda", ratio=0.70, batch=len(pairs), filter=variant)
        bm.run([(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()) for q, t in pairs])
        ok = True
        for b, (q, t) in enumerate(pairs):
            wi, wd = O.knn2(q, t, nthreads=os.cpu_count() or 8)
            wq, wt, _ = O.ratio_filter(wi, wd, 0.70)
            m = int(bm.count[b].item())
            ok = ok and np.array_equal(bm.idx[b].cpu().numpy(), wi) and np.array_equal(bm.dist[b].cpu().numpy().view(np.uint32), wd.view(np.uint32)) \
                and m == len(wq) and np.array_equal(bm.out_q[b, :m].cpu().numpy(), wq) and np.array_equal(bm.out_t[b, :m].cpu().numpy(), wt)
        st = bm.stats[0].cpu().tolist()
        modes[st[3]] = modes.get(st[3], 0) + 1
        cases += 1
        batched += 1
        if not ok:
            fails += 1
            print(f"MISMATCH case {cases}: BATCH of {len(pairs)} kind={kind} nq={nqb} nt={ntb} variant={variant} stats={st}", flush=True)
        continue
    pm = ops.PairMatcher(nq, nt, "cuda", ratio=0.70, filter=variant)
    idx, dist, oq, ot, cnt = pm.run(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda())
    gi, gd, m = idx.cpu().numpy(), dist.cpu().numpy(), int(cnt.item())
    wi, wd = O.knn2(q, t, nthreads=os.cpu_count() or 8)
    wq, wt, _ = O.ratio_filter(wi, wd, 0.70)
    ok = np.array_equal(gi, wi) and np.array_equal(gd.view(np.uint32), wd.view(np.uint32)) and m == len(wq) \
        and np.array_equal(oq[:m].cpu().numpy(), wq) and np.array_equal(ot[:m].cpu().numpy(), wt)
    st = pm.stats.cpu().tolist()
    modes[st[3]] = modes.get(st[3], 0) + 1
    cases += 1
    if not ok:
        fails += 1
        print(f"MISMATCH case {cases}: kind={kind} nq={nq} nt={nt} variant={variant} stats={st} rows differing={(gi != wi).any(1).sum()}", flush=True)
print(f"fuzz: {cases} cases ({batched} of them batches of 2..8 pairs), {fails} mismatches, filter modes used {modes} (seed {seed})")
sys.exit(1 if fails else 0)
