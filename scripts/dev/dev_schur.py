"""Dev: Schur-complement LM vs alternating block updates on a dense problem; timing of the products at config-4 scale."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from datagen import ba_problem
from sfm_mvs_amd import ba, ops
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
K, cams, X, obs = ba_problem(12, 3000, 0.5, seed=33, perturb=0.01)
c1, x1, h1 = ba.bundle_adjust_schur(cu(cams), K, cu(X), cu(obs), iters=8, log=print)
c2, x2, h2 = ba.bundle_adjust(cu(cams), K, cu(X), cu(obs), iters=8)
print("schur:", [f"{h:.5g}" for h in h1]); print("altern:", [f"{h:.5g}" for h in h2]); print("noise floor", 2 * 12 * 3000 * 0.25)
if len(sys.argv) > 1:
    ncam, npt = 500, 200000
    K, cams, X, obs = ba_problem(ncam, npt, 0.5, seed=3, perturb=0.01)
    cd, Xd = cu(cams), cu(X)
    x = torch.randn((ncam, 6), dtype=torch.float64, device="cuda"); v = torch.randn((npt, 3), dtype=torch.float64, device="cuda")
    for name, f in (("W^T x", lambda: ops.ba_schur_wt(cd, K, Xd, x)), ("W v", lambda: ops.ba_schur_w(cd, K, Xd, v))):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"{name}: {dt*1e3:.2f} ms per product at 500 x 200k = {ncam*npt/dt:.3e} pairs/s")
    od = cu(obs)
    t0 = time.perf_counter()
    c, xx, h = ba.bundle_adjust_schur(cd, K, Xd, od, iters=4, log=print)
    torch.cuda.synchronize(); print(f"4 LM iterations at config-4 scale: {time.perf_counter()-t0:.2f} s, cost {h[0]:.4g} -> {h[-1]:.4g} (floor {2*ncam*npt*0.25:.4g})")
