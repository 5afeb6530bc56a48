"""The exact-integer body of the KNN (v_mfma_i32_32x32x32_i8, csrc/knn.hip: filter_i8_body / refine_i8_body) — what
`filter="auto"` runs when every value of the batch is an integer 0..255, i.e. on cv2 SIFT output
(/root/reference sfm.py:246-252 -> knnMatch sfm.py:259-260).  Every case is compared BIT FOR BIT (indices and float32
distances) with the direct-form oracle `orc_knn2_l2_f32`, and `stats[3]` says which arithmetic ran (4 = i8)."""
import numpy as np
import pytest
import torch

from datagen import planted_pair, sift_like

pytestmark = pytest.mark.gpu


def run(hip, q, t, filter="auto"):
    out = hip.knn2(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda(), return_stats=True, filter=filter)
    torch.cuda.synchronize()
    return tuple(o.cpu().numpy() for o in out)


def check(hip, oracle, q, t, mode=4, max_rescan=None, threads=8):
    gi, gd, st = run(hip, q, t)
    wi, wd = oracle.knn2(q, t, nthreads=threads)
    assert st[3] == mode, f"filter arithmetic {st[3]}, expected {mode}"
    assert np.array_equal(gi, wi), f"{(gi != wi).any(1).sum()} rows differ"
    assert np.array_equal(gd.view(np.uint32), wd.view(np.uint32))
    if max_rescan is not None:
        assert st[0] <= max_rescan, f"{st[0]} queries rescanned"
    return st


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (5, 3), (31, 33), (64, 64), (129, 1000), (777, 1234), (1024, 8192), (1025, 8193),
                                   (3000, 2500), (6200, 700), (100, 20000), (20000, 96)])
def test_sift_like_shapes(hip, oracle, nq, nt):
    rng = np.random.default_rng(nq * 31 + nt)
    q, t = sift_like(rng, nq), sift_like(rng, nt)
    check(hip, oracle, q, t)


def test_planted_pairs_no_rescans(hip, oracle):
    """Ordinary SIFT-like data with planted matches: exact top-2, and the integer certificate never has to rescan."""
    rng = np.random.default_rng(5)
    q, t, _ = planted_pair(rng, 4000, 5000, 0.3)
    check(hip, oracle, q, t, max_rescan=4)


def test_uniform_u8_full_range(hip, oracle):
    """Arbitrary bytes 0..255 (not SIFT-shaped): d^2 up to 8.3e6, where float32 square roots of neighbouring integers collide."""
    rng = np.random.default_rng(6)
    q = rng.integers(0, 256, (1500, 128)).astype(np.float32)
    t = rng.integers(0, 256, (2600, 128)).astype(np.float32)
    check(hip, oracle, q, t)


@pytest.mark.parametrize("v", [207, 201, 255])
def test_sqrt_collisions_decide_by_index(hip, oracle, v):
    """d^2 in (2^22, 2^23): two consecutive integers can share one float32 sqrt, and the reference orders by the FLOAT distance
    (ties -> lower index).  Against an all-zero query, rows of 126 x v + two elements from {0, 1} have d^2 = C, C + 1, C + 2 with
    C = 126 v^2:  v = 207: sqrtf(C) == sqrtf(C + 1);  v = 201, 255: sqrtf(C + 1) == sqrtf(C + 2).  33 000 rows = 1 032 tiles: every
    filter workgroup owns several tiles, streams are full, nothing can be certified — the rescan path decides."""
    rng = np.random.default_rng(7 + v)
    C = 126 * v * v
    assert C > 2 ** 22 and (np.sqrt(np.float32(C)) == np.sqrt(np.float32(C + 1)) or np.sqrt(np.float32(C + 1)) == np.sqrt(np.float32(C + 2)))
    nt = 33000
    t = np.full((nt, 128), float(v), np.float32)
    t[:, 126:] = rng.integers(0, 2, (nt, 2)).astype(np.float32)
    t[:40, 126:] = 1.0                                             # the first rows are all C + 2: the winners sit further down
    t[40:80, 126] = 1.0
    t[40:80, 127] = 0.0                                            # then C + 1
    q = np.zeros((64, 128), np.float32)
    q[1::2, 5] = 1.0                                               # a second kind of query (every distance shifts a little)
    st = check(hip, oracle, q, t)
    assert st[0] == 64                                             # ties en masse: every query took the rescan path


def test_duplicates_and_all_equal_rows(hip, oracle):
    """Degenerate train sets: every row equal / blocks of duplicates — nothing can be certified, everything is rescanned."""
    rng = np.random.default_rng(8)
    base = sift_like(rng, 40)
    t = np.repeat(base, 30, axis=0)                                # 1200 rows, 30 copies each
    q = sift_like(rng, 300)
    q[:40] = base
    check(hip, oracle, q, t)
    t2 = np.tile(base[:1], (700, 1))
    check(hip, oracle, q, t2)


def test_parity_bit_near_ties(hip, oracle):
    """Rows whose d^2 differ by exactly 1 (the filter's accumulator holds floor(score / 2): the parity bit is resolved by the refine)."""
    rng = np.random.default_rng(9)
    q = sift_like(rng, 200)
    t = sift_like(rng, 3000)
    for k in range(200):                                           # plant twins at distance^2 = 4, 5, 5, 6 in random rows
        rows = rng.choice(3000, 4, replace=False)
        for n, r in enumerate(rows):
            v = q[k].copy()
            cols = rng.choice(128, 4 + (n + 1) // 2, replace=False)
            bump = np.where(v[cols] < 128, 1.0, -1.0)
            v[cols] += bump
            if n % 2 == 0:
                v[cols[0]] += bump[0]                              # one coordinate off by 2: d^2 = 4 + (len - 1) ... mixed parities
            t[r] = v
    check(hip, oracle, q, t)


def test_extreme_rows_fall_back_to_fp16(hip, oracle):
    """|t - 127|^2 / 2 spread beyond the init product's range (all-127 rows next to all-255 rows): the batch takes the fp16 body
    (mode 0: integers are exact in fp16) — same results."""
    rng = np.random.default_rng(10)
    q = sift_like(rng, 500)
    t = sift_like(rng, 800)
    t[3] = 127.0
    t[4] = 255.0
    check(hip, oracle, q, t, mode=0)
    # within range again: all-zero rows (|t - 127|^2 = 2 064 512) beside ordinary descriptors
    t[3] = 0.0
    t[4] = sift_like(rng, 1)[0]
    check(hip, oracle, q, t, mode=4)


def test_non_integer_value_falls_back(hip, oracle):
    rng = np.random.default_rng(11)
    q, t, _ = planted_pair(rng, 600, 900, 0.3)
    t[17, 5] += 0.5
    check(hip, oracle, q, t, mode=0)
    t[17, 5] = 256.0                                               # an integer, but not a u8
    check(hip, oracle, q, t, mode=0)
    t[17, 5] = -1.0
    check(hip, oracle, q, t, mode=0)
    t[17, 5] = -0.0                                                # minus zero IS zero
    check(hip, oracle, q, t, mode=4)


def test_half_variant_never_takes_the_integer_body(hip, oracle):
    rng = np.random.default_rng(12)
    q, t, _ = planted_pair(rng, 700, 1100, 0.3)
    gi, gd, st = run(hip, q, t, filter="half")
    assert st[3] == 0
    wi, wd = oracle.knn2(q, t, nthreads=8)
    assert np.array_equal(gi, wi) and np.array_equal(gd.view(np.uint32), wd.view(np.uint32))


@pytest.mark.parametrize("batch", [2, 5, 8])
def test_batched_integer_pairs(hip, oracle, batch):
    """Batches run ONE body: all pairs u8 -> i8 (mode 4); one heavy-tailed float pair in the batch -> the whole batch takes
    16-bit arithmetic (the u8 chunks are converted from the byte image); one float pair with compact support -> the batch runs
    the integer body on quantised data (mode 5), the u8 pairs on the grid s = 1."""
    rng = np.random.default_rng(13 + batch)
    nq, nt = 1500, 2100
    pairs = [planted_pair(rng, nq, nt, 0.3)[:2] for _ in range(batch)]
    dev = torch.device("cuda:0")
    bm = hip.BatchMatcher(nq, nt, dev, ratio=0.70, batch=batch)
    bm.run([(torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)) for q, t in pairs])
    torch.cuda.synchronize()
    for b, (q, t) in enumerate(pairs):
        wi, wd = oracle.knn2(q, t, nthreads=8)
        assert int(bm.stats[b, 3]) == 4
        assert np.array_equal(bm.idx[b].cpu().numpy(), wi) and np.array_equal(bm.dist[b].cpu().numpy().view(np.uint32), wd.view(np.uint32))
        wq, wt, _ = oracle.ratio_filter(wi, wd, 0.70)
        m = int(bm.count[b].item())
        assert m == len(wq) and np.array_equal(bm.out_q[b, :m].cpu().numpy(), wq) and np.array_equal(bm.out_t[b, :m].cpu().numpy(), wt)
    for other in (((rng.standard_normal((nq, 128)) * 30).astype(np.float32), (rng.standard_normal((nt, 128)) * 30).astype(np.float32), 1),
                        (rng.random((nq, 128), dtype=np.float32), rng.random((nt, 128), dtype=np.float32), 5)):
        pairs[batch // 2] = other[:2]
        mode = other[2]
        bm.run([(torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)) for q, t in pairs])
        torch.cuda.synchronize()
        for b, (q, t) in enumerate(pairs):
            wi, wd = oracle.knn2(q, t, nthreads=8)
            assert int(bm.stats[b, 3]) == mode
            assert np.array_equal(bm.idx[b].cpu().numpy(), wi) and np.array_equal(bm.dist[b].cpu().numpy().view(np.uint32), wd.view(np.uint32))


def test_config2_size_sift_like(hip, oracle):
    """BASELINE configs[1] shape (10k x 10k) on SIFT-like integers with 30 % planted matches (SURVEY 8d distribution (ii))."""
    rng = np.random.default_rng(14)
    q, t, _ = planted_pair(rng, 10000, 10000, 0.3)
    check(hip, oracle, q, t, max_rescan=8, threads=64)
