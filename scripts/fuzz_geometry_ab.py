"""scripts/fuzz_geometry.py with sfm_solve_pnp_ransac run through BOTH of its paths — the resident PnP server and a launch + stream
synchronisation per step (sfm_debug_pnp_sweep_server) — on every case: the two must agree bit for bit (ok, rvec, tvec, inlier list,
info), whatever the oracle says.  Same arguments as fuzz_geometry.py (the random sequence is the same, so a seed reproduces its cases)."""
import os, sys, runpy
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sfm_mvs_amd import ransac, _lib
L = _lib.lib()
_orig = ransac.solve_pnp_ransac
stats = {"calls": 0, "differ": 0}


def both_paths(*a, **k):
    got = _orig(*a, **k)
    L.sfm_debug_pnp_sweep_server(0)
    try:
        ref = _orig(*a, **k)
    finally:
        L.sfm_debug_pnp_sweep_server(1)
    stats["calls"] += 1
    same = bool(got[0]) == bool(ref[0])
    if same and got[0]:
        for x, y in zip(got[1:4], ref[1:4]):
            x, y = np.asarray(x.cpu() if hasattr(x, "cpu") else x), np.asarray(y.cpu() if hasattr(y, "cpu") else y)
            same = same and x.shape == y.shape and np.array_equal(np.isnan(x.astype(float)), np.isnan(y.astype(float))) and np.array_equal(x[~np.isnan(x.astype(float))], y[~np.isnan(y.astype(float))])
        if len(got) > 4:
            same = same and list(got[4]) == list(ref[4])
    if not same:
        stats["differ"] += 1
        n = len(a[0])
        print(f"SERVER != LAUNCH PATH: n {n} info {list(got[4]) if len(got) > 4 else None} vs {list(ref[4]) if len(ref) > 4 else None}; rvec {np.ravel(got[1]) if got[0] else None} vs {np.ravel(ref[1]) if ref[0] else None}")
    return got


ransac.solve_pnp_ransac = both_paths
try:
    runpy.run_path(os.path.join(ROOT, "scripts", "fuzz_geometry.py"), run_name="__main__")
finally:
    print(f"fuzz_geometry_ab: {stats['calls']} solvePnPRansac calls through both paths, {stats['differ']} differ")
