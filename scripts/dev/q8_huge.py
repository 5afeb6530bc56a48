import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/sfm_mvs_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
from sfm_mvs_amd import ops
from oracle import oracle as O
O.lib()
rng = np.random.default_rng(3)
for scale in (1e12, 1e17, 3e18, 1e19, 1e20, 1e30):
    for filt in ("auto", "noquant"):
        q = (rng.random((300, 128)) * scale).astype(np.float32); t = (rng.random((700, 128)) * scale).astype(np.float32)
        gi, gd, st = ops.knn2(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda(), return_stats=True, filter=filt)
        torch.cuda.synchronize()
        wi, wd = O.knn2(q, t, nthreads=16)
        gi, gd = gi.cpu().numpy(), gd.cpu().numpy()
        print(f"scale {scale:g} {filt:8s} mode {st.cpu().numpy()[3]} idx rows differ {(gi != wi).any(1).sum()} dist rows differ {(gd.view(np.uint32) != wd.view(np.uint32)).any(1).sum()} inf dists {np.isinf(wd).sum()}", flush=True)
