"""Pins the oracle's KNN / ratio restatement (sfm.py:259-265) with known-answer cases."""
import numpy as np

from datagen import planted_pair, sift_like


def brute_f64(q, t):
    d = np.sqrt(((q[:, None, :].astype(np.float64) - t[None].astype(np.float64)) ** 2).sum(-1))
    o = np.argsort(d, axis=1, kind="stable")[:, :2]
    return o, np.take_along_axis(d, o, 1)


def test_matches_float64_bruteforce_on_random_floats(oracle):
    rng = np.random.default_rng(1)
    q, t = rng.random((300, 128), dtype=np.float32), rng.random((500, 128), dtype=np.float32)
    idx, dist = oracle.knn2(q, t)
    o, d = brute_f64(q, t)
    assert np.array_equal(idx, o)
    assert np.abs(dist - d).max() <= 1e-6 * d.max()


def test_exact_on_integer_sift_like_data_and_planted_matches(oracle):
    rng = np.random.default_rng(2)
    q, t, planted = planted_pair(rng, 400, 600, 0.3)
    idx, dist = oracle.knn2(q, t)
    o, d = brute_f64(q, t)
    # integer-valued data: every float32 sum is exact → the distance is the correctly rounded sqrt
    assert np.array_equal(dist, np.sqrt(np.round(d ** 2)).astype(np.float32))
    assert np.array_equal(idx[:, 0], o[:, 0])
    assert (idx[planted[:, 0], 0] == planted[:, 1]).mean() > 0.98
    oq, ot, mask = oracle.ratio_filter(idx, dist, 0.70)
    assert mask[planted[:, 0]].mean() > 0.9 and mask.sum() <= len(planted) + 5
    assert np.array_equal(oq, np.flatnonzero(mask)) and np.array_equal(ot, idx[mask.astype(bool), 0])


def test_ties_keep_the_lower_train_index(oracle):
    rng = np.random.default_rng(3)
    base = sift_like(rng, 10)
    t = np.vstack([base, base, base])          # every train row appears 3 times
    q = base[:4].copy()
    idx, dist = oracle.knn2(q, t)
    assert np.array_equal(idx[:, 0], np.arange(4)) and np.array_equal(idx[:, 1], np.arange(4) + 10)
    assert np.all(dist == 0)


def test_ratio_is_strict_and_in_double(oracle):
    idx = np.array([[0, 1], [2, 3], [4, -1]], np.int32)
    d2 = np.float32(10.0)
    dist = np.array([[np.float32(7.0), d2], [np.nextafter(np.float32(7.0), np.float32(0)), d2], [1.0, np.inf]], np.float32)
    oq, ot, mask = oracle.ratio_filter(idx, dist, 0.70)
    # 7.0 < 0.7*10.0 is False in double (0.7*10 = 7.000000000000001 → True!); mirror Python exactly
    expect = [float(dist[i, 0]) < 0.70 * float(dist[i, 1]) for i in range(2)] + [False]
    assert mask.tolist() == [int(e) for e in expect]


def test_fewer_than_two_trains(oracle):
    q = np.ones((3, 128), np.float32)
    idx, dist = oracle.knn2(q, np.zeros((1, 128), np.float32))
    assert idx.tolist() == [[0, -1]] * 3 and np.all(np.isinf(dist[:, 1]))


def test_threads_do_not_change_results(oracle):
    rng = np.random.default_rng(4)
    q, t = rng.random((257, 128), dtype=np.float32), rng.random((300, 128), dtype=np.float32)
    a = oracle.knn2(q, t, nthreads=1)
    b = oracle.knn2(q, t, nthreads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
