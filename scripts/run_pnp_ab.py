"""A/B of sfm_solve_pnp_ransac's Levenberg-Marquardt loop: resident sweep server (round 6) vs a launch + stream synchronisation
per sweep (round 5), same box, same scenes.  Prints per-call host time and the library's own breakdown (sfm_pnp_profile_read)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from datagen import gustav_pair
from sfm_mvs_amd import ransac, _lib
L = _lib.lib()
NAMES = ("copy_in", "epnp_host", "score+wait", "mask/inliers", "dlt_init", "lm_sweeps+wait", "lm_algebra")
import ctypes
def prof(reset=True):
    out = (ctypes.c_double * 10)()
    L.sfm_pnp_profile_read(out, 1 if reset else 0)
    return list(out)
for n, n_bad in ((300, 30), (800, 80), (2000, 200), (6000, 600)):
    K, P1, P2, X, x1, x2 = gustav_pair(7, n, 0.3, seed=n)
    rng = np.random.default_rng(n)
    bad = rng.permutation(n)[:n_bad]
    x2 = x2.copy(); x2[bad] += rng.uniform(10, 150, (n_bad, 2)).astype(np.float32)
    Xd, ud = torch.from_numpy(X.astype(np.float32)).cuda(), torch.from_numpy(x2).cuda()
    res = {}
    for mode in (1, 0, 1, 0):
        L.sfm_debug_pnp_sweep_server(mode)
        for _ in range(5):
            ransac.solve_pnp_ransac(Xd, ud, K, return_device_inliers=True)
        torch.cuda.synchronize(); prof()
        s0, p0 = L.sfm_host_sync_count(), L.sfm_host_poll_count()
        t0 = time.perf_counter()
        reps = 200
        for _ in range(reps):
            r = ransac.solve_pnp_ransac(Xd, ud, K, return_device_inliers=True, want_info=True)
        dt = (time.perf_counter() - t0) / reps * 1e6
        pr = prof()
        res[mode] = r
        print(f"n={n:5d} inliers={int(r[4][1]):5d} {'server' if mode else 'launch'}: {dt:7.1f} us/call  syncs/call {(L.sfm_host_sync_count() - s0) / reps:5.2f}  polls/call {(L.sfm_host_poll_count() - p0) / reps:8.0f}  "
              + "  ".join(f"{k} {pr[1 + i] / pr[0]:.1f}" for i, k in enumerate(NAMES)) + f"  sweeps/call {pr[9] / pr[0]:.2f}")
    L.sfm_debug_pnp_sweep_server(1)
    a, b = res[1], res[0]
    print("   identical:", bool(np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and torch.equal(a[3], b[3]) and list(a[4]) == list(b[4])))
