"""Randomised parity sweep of the KNN path (sfm_match_l2_f32 / sfm_match_batch_l2_f32 through ops.PairMatcher /
ops.BatchMatcher) against the CPU oracle: indices, float32 distances (bit patterns), Lowe survivors.

  python scripts/fuzz_knn.py [seconds] [seed] [big | q8]

Data families: the ordinary ones (uniform, scaled normals, SIFT-like integers, planted twins, duplicates, near-ties, unit
vectors, mixed magnitudes) and a second group aimed at the margins of the exactness certificate (docs/knn.md):
  cancel     operands that maximise |c| + sum |a b| while the score itself cancels to almost nothing (alternating signs,
             large common offset): the regime in which the matrix pipe's accumulation error is largest (7.1 units measured)
  tie23      every query has its 2nd and 3rd neighbour at distances that differ by 0 .. a few float32 ulps (and the 1st / 2nd
             likewise for a third of them): a mis-ranked pair flips an index or a Lowe decision
  pow2       squared distances within a few ulps of a power of two (sqrtf / key-truncation boundaries)
  fp16edge   values on fp16 rounding midpoints, at the top of its range (~6e4) and around its smallest normals (6.1e-5)
  normspread query norms spread over 10^3 inside one pair (the q4 filter's scores carry the pair's LARGEST ||q||^2 and its slack is
             relative to that: small-norm queries of such a pair rescan more, and must stay exact)
and a third group of u8-integer data — what cv2 SIFT emits and what filter="auto" sends through the exact-integer i8 body:
  u8_uniform arbitrary bytes (ranges 2 .. 256);  u8_ties  trains one or two units apart (parity bit, exact ties inside a record);
  u8_far     d^2 in (2^22, 2^23): float32 square roots of neighbouring integers collide;  u8_extreme  rows at the ends of the init
  product's range (all-0 / all-127 / all-255 / four saturated bins: some pairs fall back to the 16-bit body);  u8_dups  duplicates.
and a fourth group of float data with COMPACT support — what filter="auto" QUANTISES to 8 bits for the integer body (stats[3] = 5):
  q8_uniform uniform values on a random interval (offsets / scales over six decades, either sign);  q8_beta  bounded, non-uniform;
  q8_twins   planted near-twins and exact duplicates (d far below the quantisation step: the slack dominates the distance);
  q8_grid    values ON a 256-level grid (residuals ~ 0: exact ties of the quantised distances, the float32 order decides);
  q8_clip    outliers in rows the grid's sample does not see (clipped: the measured residuals reject the grid, the pair is repaired);
  q8_const   constant rows at the middle and the ends of the range (init product's range exceeded: repair).
Batches mix the families: a u8 pair next to a float pair exercises the conversion of byte-image chunks to fp16 (mixed batch),
a quantised pair next to a Gaussian one the repair of pairs quantised in vain.
Most cases are small (the oracle dominates the wall time); one in eight is large, `big` adds 20k-70k train rows.
The log ends with the sha256 of csrc/knn.hip and of its CODE (comments and whitespace removed, scripts/knn_code_hash.py):
the round's logs under profiles/ are checked against the LOADED binary's code hash (sfm_build_id) by tests/test_gpu_knn.py.
"""
import hashlib
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")      # (the oracle's OpenMP team and torch's must not spin against each other)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sfm_mvs_amd import ops
from oracle import oracle as O
from datagen import planted_pair, sift_like

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = len(sys.argv) > 3 and sys.argv[3] == "big"       # also long train sets (many substreams / candidate records per query)
q8only = len(sys.argv) > 3 and sys.argv[3] == "q8"     # only the families the quantised integer body runs (and what sits next to them in a batch)
rng = np.random.default_rng(seed)
NTH = os.cpu_count() or 8
f32 = np.float32


def unit_rows(n):
    v = rng.standard_normal((n, 128))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def make(kind, nq, nt):
    if kind == "uniform":
        return rng.random((nq, 128), dtype=f32), rng.random((nt, 128), dtype=f32)
    if kind == "normal_scaled":
        s = f32(10.0 ** rng.uniform(-4, 4))
        return (rng.standard_normal((nq, 128)) * s).astype(f32), (rng.standard_normal((nt, 128)) * s + s).astype(f32)
    if kind == "sift":
        return sift_like(rng, nq), sift_like(rng, nt)
    if kind == "planted":
        q, t, _ = planted_pair(rng, nq, max(nt, 2), 0.3)
        return q, t
    if kind == "duplicates":
        base = rng.random((max(1, nt // 7), 128), dtype=f32)
        t = np.tile(base, (8, 1))[:nt]
        q = base[rng.integers(0, len(base), nq)] + f32(1e-3) * rng.standard_normal((nq, 128)).astype(f32)
        return q.astype(f32), t
    if kind == "near_ties":
        base = rng.random((1, 128), dtype=f32)
        t = (base * (1 + f32(1e-6) * rng.standard_normal((nt, 1)).astype(f32))).astype(f32)
        return rng.random((nq, 128), dtype=f32), t
    if kind == "unit":
        return unit_rows(nq).astype(f32), unit_rows(nt).astype(f32)
    if kind == "mixed_magnitude":
        q = rng.random((nq, 128), dtype=f32); t = rng.random((nt, 128), dtype=f32)
        t[:: max(1, nt // 9)] *= f32(1e-4); q[::3] *= f32(100.0)
        return q, t
    # ---- aimed at the certificate's margins
    if kind == "cancel":
        # alternating-sign pattern on a large common offset: |q.t| terms are ~offset^2 each and cancel pairwise
        off = 10.0 ** rng.uniform(0, 3)
        sign = np.where(np.arange(128) % 2 == 0, 1.0, -1.0)
        q = (off * sign + rng.standard_normal((nq, 128))) * rng.choice([1.0, -1.0], (nq, 1))
        t = (off * sign[::-1] * rng.choice([1.0, -1.0], (nt, 1)) + rng.standard_normal((nt, 128)))
        return q.astype(f32), t.astype(f32)
    if kind in ("tie23", "pow2"):
        nt = max(nt, 8)
        q = (rng.random((nq, 128)) * 10.0 ** rng.uniform(-1, 2)).astype(f32)
        t = (rng.random((nt, 128)) * 10.0 ** rng.uniform(-1, 2)).astype(f32)
        m = min(nq, nt // 3)
        rows = rng.permutation(nt)[: 3 * m].reshape(m, 3)
        u, v, w = unit_rows(m), unit_rows(m), unit_rows(m)
        for i in range(m):
            qi = q[i].astype(np.float64)
            r = (0.02 + 0.2 * rng.random()) * (np.linalg.norm(qi) + 1e-3)
            if kind == "pow2":
                r = np.sqrt(2.0 ** np.round(np.log2(r * r)))                 # d^2 at a power of two (up to rounding of t)
            eta = rng.choice([0.0, 1e-7, 3e-7, 1e-6, 1e-5]) * rng.choice([1, -1])
            r1 = r * (1 + (rng.choice([0.0, 1e-7, 1e-6]) if i % 3 == 0 else -0.2))
            t[rows[i, 0]] = (qi + r1 * u[i]).astype(f32)
            t[rows[i, 1]] = (qi + r * v[i]).astype(f32)
            t[rows[i, 2]] = (qi + r * (1 + eta) * w[i]).astype(f32)
        return q, t
    if kind == "fp16edge":
        pick = rng.integers(0, 3)
        if pick == 0:      # midpoints between adjacent fp16 values (ties of the round-to-nearest-even)
            base = (rng.integers(1024, 2048, (nq + nt, 128)).astype(np.float64) + 0.5) * 2.0 ** rng.integers(-12, 4)
        elif pick == 1:    # top of fp16's range: |-2 q| must stay <= 60000
            base = rng.uniform(2.0e4, 2.9e4, (nq + nt, 128)) * rng.choice([1.0, -1.0], (nq + nt, 128))
        else:              # around fp16's smallest normals (6.1e-5): the matrix pipe may flush what is below
            base = rng.uniform(2e-5, 2.5e-4, (nq + nt, 128)) * rng.choice([1.0, -1.0], (nq + nt, 128)) + (rng.random((nq + nt, 128)) < 0.02) * 0.7
        return base[:nq].astype(f32), base[nq:].astype(f32)
    # ---- u8 integers: the exact-integer body (v_mfma_i32_32x32x32_i8; filter="auto" picks it on the device) and its margins
    if kind == "u8_uniform":
        hi_q, hi_t = int(rng.choice([2, 16, 120, 256])), int(rng.choice([2, 16, 120, 256]))
        return rng.integers(0, hi_q, (nq, 128)).astype(f32), rng.integers(0, hi_t, (nt, 128)).astype(f32)
    if kind == "u8_ties":
        # trains = one base row with a few +-1 tweaks: squared distances to any query differ by 0, 1, 2, ... (the filter's
        # accumulator holds floor(score / 2): the parity bit and exact ties are the refine kernel's), many per 8-row record
        base = sift_like(rng, 1)[0] if rng.random() < 0.5 else rng.integers(0, 256, 128).astype(f32)
        t = np.tile(base, (nt, 1))
        for r in range(nt):
            cols = rng.choice(128, int(rng.integers(0, 4)), replace=False)
            t[r, cols] = np.clip(t[r, cols] + rng.choice([-1.0, 1.0], len(cols)), 0, 255)
        q = np.clip(np.tile(base, (nq, 1)) + np.rint(rng.normal(0, rng.choice([0.0, 1.0, 20.0]), (nq, 128))), 0, 255).astype(f32)
        return q, t.astype(f32)
    if kind == "u8_far":
        # queries near 0, trains near 255: d^2 in (2^22, 2^23), where two consecutive integers can share a float32 square root and
        # the reference's float compare (ties -> lower index) decides
        v = int(rng.integers(182, 256))
        t = np.full((nt, 128), float(v), f32)
        k = int(rng.integers(1, 4))
        t[:, :k] = rng.integers(0, 3, (nt, k)).astype(f32)
        q = np.zeros((nq, 128), f32)
        q[:, 100:] = rng.integers(0, 2, (nq, 28)).astype(f32)
        return q, t
    if kind == "u8_extreme":
        # rows at the ends of the init product's range: all-zero, all-127, all-255, four saturated bins; with both ends present
        # the pair falls back to the 16-bit body (stats[3] = 0) — results must not change
        q, t = sift_like(rng, nq), sift_like(rng, nt)
        for arr in (q, t):
            for r in rng.integers(0, len(arr), max(1, len(arr) // 50)):
                pick = rng.integers(0, 5)
                if pick == 0: arr[r] = 0.0
                elif pick == 1: arr[r] = 255.0 if rng.random() < 0.5 else 254.0
                elif pick == 2: arr[r] = 127.0
                elif pick == 3:
                    arr[r] = 0.0
                    arr[r, rng.choice(128, 4, replace=False)] = 255.0
                else: arr[r] = float(rng.integers(0, 256))
        if rng.random() < 0.5:
            t[t == 127.0] = 126.0                                 # (keep floor(w / 2) inside the range: the integer body runs)
        return q, t
    if kind == "u8_dups":
        base = sift_like(rng, max(1, nt // 9))
        t = np.tile(base, (10, 1))[:nt]
        q = np.clip(base[rng.integers(0, len(base), nq)] + np.rint(rng.normal(0, 1.0, (nq, 128))), 0, 255).astype(f32)
        return q, t.astype(f32)
    if kind in ("q8_uniform", "q8_beta", "q8_twins", "q8_grid", "q8_clip", "q8_const"):
        lo = float(rng.choice([0.0, 0.0, -1.0, 1.0]) * 10.0 ** rng.uniform(-3, 3))
        width = float(10.0 ** rng.uniform(-3, 3))
        if abs(lo) > 300 * width:
            lo = 0.0                                               # (keep the range well above the values' float32 spacing)
        draw = (lambda n: rng.beta(2.0, 2.0, (n, 128))) if kind == "q8_beta" else (lambda n: rng.random((n, 128)))
        if kind == "q8_grid":
            draw = lambda n: rng.integers(0, 256, (n, 128)) / 255.0
        q, t = lo + width * draw(nq), lo + width * draw(nt)
        if kind == "q8_grid":
            t[0], q[0] = lo, lo + width                            # (the sample sees both ends)
        if kind == "q8_twins":
            m = min(nq, nt) // 2
            rows = rng.permutation(nt)[:m]
            noise = rng.choice([0.0, 1e-7, 1e-5, 1e-3]) * width
            t[rows] = q[rng.permutation(nq)[:m]] + noise * rng.standard_normal((m, 128))
            if m > 2: t[rows[1]] = t[rows[0]]                      # an exact duplicate
        if kind == "q8_clip" and nt > 40:
            rows = np.setdiff1d(np.arange(nt), (np.arange(16) * nt) >> 4)      # not the rows the sample reads
            hit = rng.choice(rows, max(1, len(rows) // int(rng.choice([3, 50, 500]))), replace=False)
            t[hit, rng.integers(0, 128, len(hit))] = lo + width * rng.choice([3.0, -2.0, 40.0])
        if kind == "q8_const" and nt > 8:
            t[1], t[2], t[3] = lo + 0.5 * width, lo, lo + width
            if nq > 4: q[1] = lo + 0.5 * width
        return q.astype(f32), t.astype(f32)
    if kind == "normspread":
        q = rng.random((nq, 128)) * 10.0 ** rng.uniform(-1.5, 1.5, (nq, 1))
        t = rng.random((nt, 128)) * 10.0 ** rng.uniform(-1.5, 1.5, (nt, 1))
        return q.astype(f32), t.astype(f32)
    raise ValueError(kind)


kinds = ["uniform", "normal_scaled", "sift", "planted", "duplicates", "near_ties", "unit", "mixed_magnitude",
         "cancel", "tie23", "pow2", "fp16edge", "normspread", "tie23", "cancel",
         "u8_uniform", "u8_ties", "u8_far", "u8_extreme", "u8_dups", "planted", "u8_ties",
         "q8_uniform", "q8_beta", "q8_twins", "q8_grid", "q8_clip", "q8_const", "q8_uniform", "q8_twins"]
variants = ["auto"] * 8 + ["half", "split", "f32", "lds", "lds_split", "noquant"]
if q8only:
    kinds = ["q8_uniform", "q8_beta", "q8_twins", "q8_grid", "q8_clip", "q8_const", "uniform", "q8_uniform", "sift", "q8_twins", "normal_scaled"]
    variants = ["auto"] * 6 + ["noquant"]
t_start = time.time()
t_end = t_start + budget
cases = fails = batched = 0
modes, per_kind, per_variant, rescans = {}, {}, {}, 0


def check(q, t, idx, dist, oq, ot, m):
    wi, wd = O.knn2(q, t, nthreads=max(1, min(NTH // 2, len(q) * len(t) // 400000)))     # (a 256-thread team costs more than a small case)
    wq, wt, _ = O.ratio_filter(wi, wd, 0.70)
    return (np.array_equal(idx, wi) and np.array_equal(dist.view(np.uint32), wd.view(np.uint32)) and m == len(wq)
            and np.array_equal(oq[:m], wq) and np.array_equal(ot[:m], wt))


while time.time() < t_end:
    kind = kinds[cases % len(kinds)]
    variant = variants[(cases // 3) % len(variants)]
    if cases % 8 == 7:       # the large shapes (several row blocks, many substreams)
        nq = int(rng.choice([1000, 2049, 5000]) if rng.random() < 0.5 else rng.integers(600, 6000))
        nt = int(rng.choice([4097, 9000]) if rng.random() < 0.5 else rng.integers(3000, 12000))
    else:
        nq = int(rng.choice([1, 3, 17, 64, 255, 257, 511, 513]) if rng.random() < 0.4 else rng.integers(1, 600))
        nt = int(rng.choice([1, 2, 31, 33, 512, 1023, 2049]) if rng.random() < 0.4 else rng.integers(1, 2600))
    if big and cases % 24 == 5:
        nq, nt = int(rng.integers(1, 700)), int(rng.integers(20000, 70000))
    if kind in ("duplicates", "near_ties", "u8_dups", "u8_ties", "u8_far", "q8_grid"):
        nq, nt = min(nq, 600), min(nt, 3000)             # every stream is rescanned: keep the exact work bounded
    q, t = make(kind, nq, nt)
    nq, nt = len(q), len(t)
    per_kind[kind] = per_kind.get(kind, 0) + 1
    per_variant[variant] = per_variant.get(variant, 0) + 1
    if variant != "f32" and cases % 4 == 1:
        # a batch of 2..8 pairs of this shape in ONE launch set (sfm_match_batch_l2_f32), data families mixed: the batch runs
        # the most general arithmetic mode any pair needs, every pair must still equal the oracle
        B = int(rng.integers(2, 9))
        nqb, ntb = min(nq, 1500), min(nt, 4000)
        pairs = [make(kinds[(cases + 3 * b) % len(kinds)] if b else kind, nqb, ntb) for b in range(B)]
        pairs = [(a, b) for a, b in pairs if len(a) >= nqb and len(b) >= ntb]
        nqb, ntb = min(len(p[0]) for p in pairs), min(len(p[1]) for p in pairs)
        pairs = [(np.ascontiguousarray(a[:nqb]), np.ascontiguousarray(b[:ntb])) for a, b in pairs]
        bm = ops.BatchMatcher(nqb, ntb, "cuda", ratio=0.70, batch=len(pairs), filter=variant)
        bm.run([(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()) for a, b in pairs])
        ok = True
        idx, dist, oq, ot, cnt = bm.idx.cpu().numpy(), bm.dist.cpu().numpy(), bm.out_q.cpu().numpy(), bm.out_t.cpu().numpy(), bm.count.cpu().numpy()
        for b, (a, c) in enumerate(pairs):
            ok = ok and check(a, c, idx[b], dist[b], oq[b], ot[b], int(cnt[b, 0]))
        st = bm.stats[0].cpu().tolist()
        batched += 1
        desc = f"BATCH of {len(pairs)} kind={kind} nq={nqb} nt={ntb}"
    else:
        pm = ops.PairMatcher(nq, nt, "cuda", ratio=0.70, filter=variant)
        idx, dist, oq, ot, cnt = pm.run(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda())
        ok = check(q, t, idx.cpu().numpy(), dist.cpu().numpy(), oq.cpu().numpy(), ot.cpu().numpy(), int(cnt.item()))
        st = pm.stats.cpu().tolist()
        desc = f"kind={kind} nq={nq} nt={nt}"
    modes[st[3]] = modes.get(st[3], 0) + 1
    rescans += st[0]
    cases += 1
    if not ok:
        fails += 1
        print(f"MISMATCH case {cases}: {desc} variant={variant} stats={st}", flush=True)
    if cases % 2000 == 0:
        print(f"  ... {cases} cases, {fails} mismatches, {time.time() - t_start:.0f} s", flush=True)

worst, scale = ops.knn_mfma_selftest_result()
sha = hashlib.sha256(open(os.path.join(ROOT, "sfm_mvs_amd", "csrc", "knn.hip"), "rb").read()).hexdigest()
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from knn_code_hash import knn_code_hash                  # comments / whitespace removed: a documentation-only edit keeps the logs valid
from sfm_mvs_amd import _lib as _sfm_lib
if _sfm_lib.knn_code_hash_of_binary() != knn_code_hash():      # the sweep ran on the LOADED binary: its id is what the log must name
    print(f"fuzz: the loaded library ({_sfm_lib.build_id()}) was not built from this csrc/knn.hip ({knn_code_hash()}): rebuild first")
    sys.exit(2)
print(f"fuzz: {cases} cases ({batched} of them batches of 2..8 pairs), {fails} mismatches, seed {seed}, {time.time() - t_start:.0f} s")
print(f"  filter arithmetic modes that ran {dict(sorted(modes.items()))}; variants {per_variant}; rescanned queries in total {rescans}")
print(f"  families {per_kind}")
print(f"  runtime MFMA self-test on this device: worst E = {worst:.2f} units, chain scale {scale}")
print(f"knn_hip_sha256 {sha}")
print(f"knn_hip_code_sha256 {_sfm_lib.knn_code_hash_of_binary()}")
print(f"sfm_build_id {_sfm_lib.build_id()}")
sys.exit(1 if fails else 0)
