"""Dev: distribution of |inverse-iteration vector - Jacobi vector| and of its estimate `sens` (triangulate.hip), per geometry."""
# (needs a dev build of the library: make -C sfm_mvs_amd/csrc CXXFLAGS+=-DSFM_DEV_BUILD — release builds reject normalise_w = 4 and ignore SFM_TRI_*)
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from sfm_mvs_amd import ops
from datagen import load_pose_csv
K, P = load_pose_csv()
n = 400_000
rng = np.random.default_rng(5)
X = np.stack([rng.uniform(-6.3, 3.6, n), rng.uniform(-2.6, 5.0, n), rng.uniform(3.2, 13.0, n)], 1)
Xh = np.c_[X, np.ones(n)].T
def obs(Pm, s):
    x = Pm @ Xh
    return torch.from_numpy(((x[:2] / x[2]).T + rng.normal(0, s, (n, 2))).astype(np.float32)).cuda().t()
for name, Pa, Pb in [("pose1-2", P[1], P[2]), ("pose10-40", P[10], P[40])] + [(f"baseline {b:g}", P[1], None) for b in (1.0, 1e-2, 1e-4)]:
    if Pb is None:
        b = float(name.split()[1]); Pb = P[1].copy(); Pb[:, 3] += P[1][:, :3] @ np.array([b, 0.0, 0.0])
    for s in (0.0, 0.3):
        r = ops.triangulate(Pa, Pb, obs(Pa, s), obs(Pb, s), normalise_w=4).cpu().numpy()
        d, sens = r[0].astype(np.float64), r[1].astype(np.float64)
        ok = d >= 0
        ratio = d[ok] / np.maximum(sens[ok], 1e-300)
        pct = lambda a: " ".join(f"{np.percentile(a, q):.1e}" for q in (50, 99, 99.99, 100))
        print(f"{name:14s} s={s}: converged {ok.mean():.3f} | diff p50/p99/p99.99/max {pct(d[ok])} | sens {pct(sens[ok])} | diff/sens {pct(ratio)}")
