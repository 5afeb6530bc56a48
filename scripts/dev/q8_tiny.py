"""Dev: tiny magnitudes (squared differences that underflow) on the auto / noquant paths against the oracle."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sfm_mvs_amd import ops
from oracle import oracle as O
O.lib()
rng = np.random.default_rng(4)
for scale in (1e-6, 1e-10, 1e-15, 1e-19, 1e-22, 1e-25, 1e-30, 1e-38):
    for filt in ("auto", "noquant"):
        q = (rng.random((300, 128)) * scale).astype(np.float32); t = (rng.random((700, 128)) * scale).astype(np.float32)
        gi, gd, st = ops.knn2(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda(), return_stats=True, filter=filt)
        torch.cuda.synchronize()
        wi, wd = O.knn2(q, t, nthreads=16)
        gi, gd = gi.cpu().numpy(), gd.cpu().numpy()
        print(f"scale {scale:g} {filt:8s} mode {st.cpu().numpy()[3]} rescans {st.cpu().numpy()[0]} idx rows differ {(gi != wi).any(1).sum()} dist rows differ {(gd.view(np.uint32) != wd.view(np.uint32)).any(1).sum()} zero dists {(wd == 0).sum()}", flush=True)
