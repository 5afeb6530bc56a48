"""Oracle vs vectors produced by EXECUTING the reference's pure-NumPy functions
(tests/golden/make_golden.py): common_points (sfm.py:215-239) and to_ply's filter (sfm.py:169-181)."""
import os

import numpy as np

from datagen import GOLDEN


def test_common_points_golden(oracle):
    z = np.load(os.path.join(GOLDEN, "common_points.npz"))
    for name in "abcd":
        i1, i2, t1, t2 = oracle.common_points(z[f"{name}_pts1"], z[f"{name}_pts2"], z[f"{name}_pts3"])
        assert np.array_equal(i1, z[f"{name}_indx1"]) and np.array_equal(i2, z[f"{name}_indx2"]), name
        assert np.array_equal(t1, z[f"{name}_temp1"].reshape(-1, 2)) and np.array_equal(t2, z[f"{name}_temp2"].reshape(-1, 2))
    # the x-OR-y quirk is exercised: case a associates rows that agree in one coordinate only
    a1, a2 = z["a_pts1"], z["a_pts2"]
    pairs = list(zip(z["a_indx1"], z["a_indx2"]))
    assert any((a1[i] != a2[j]).any() for i, j in pairs)


def test_to_ply_filter_golden(oracle):
    z = np.load(os.path.join(GOLDEN, "to_ply.npz"))
    scaled, keep = oracle.to_ply_filter(z["points"])
    lines = str(z["ply_text"]).split("\n")
    start = next(i for i, l in enumerate(lines) if "end_header" in l) + 1
    body = [l.split() for l in lines[start:] if l.strip()]
    assert int(lines[2].split()[-1]) == keep.sum() == len(body)
    got = np.array([[float(v) for v in r[:3]] for r in body])
    assert np.allclose(got, scaled[keep], atol=5e-7)          # '%f' keeps 6 decimals
    cols = z["colors"][keep]
    assert np.array_equal(np.array([[int(v) for v in r[3:]] for r in body]), cols.astype(int))
    assert keep.sum() < len(keep)                              # the far outliers were dropped
