"""Dev diagnostics: per-workgroup phase timestamps of knn_refine_kernel (and the filter's end) for one step."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sfm_mvs_amd import ops, _lib
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
kind = sys.argv[3] if len(sys.argv) > 3 else "uniform"
if kind == "sift":
    q = torch.randint(0, 120, (nq, 128), generator=torch.Generator().manual_seed(0)).float().cuda()
    t = torch.randint(0, 120, (nt, 128), generator=torch.Generator().manual_seed(1)).float().cuda()
else:
    q = torch.rand((nq, 128), generator=torch.Generator().manual_seed(0)).cuda()
    t = torch.rand((nt, 128), generator=torch.Generator().manual_seed(1)).cuda()
pm = ops.PairMatcher(nq, nt, q.device)
for _ in range(3): pm.run(q, t)
nwg = (nq + 15) // 16
tr = torch.zeros(16384 + 16 * nwg + 64, dtype=torch.int64, device="cuda")
_lib.lib().sfm_debug_set_trace(ctypes.c_void_p(tr.data_ptr()))
pm.run(q, t); torch.cuda.synchronize()
_lib.lib().sfm_debug_set_trace(None)
G = int(pm.stats[1].item())
f = tr[:4 * G].view(G, 4).cpu().numpy()
a = tr[16384:16384 + 16 * nwg].view(nwg, 16).cpu().numpy()
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
print("filter end -> first refine start: %.1f us" % us(f[:, 1].max()).item() if False else "filter last end at %.1f us (refine first start = 0)" % ((f[:, 1].max() - t0) / 100.0))
st, e1, e2, e3, en = (us(a[:, k]) for k in range(5))
print(f"refine WGs {nwg}: start min {st.min():.1f} med {np.median(st):.1f} max {st.max():.1f} | end min {en.min():.1f} med {np.median(en):.1f} max {en.max():.1f}")
for name, d in (("Q rows->LDS", e1 - st), ("sweep1", e2 - e1), ("sweep2", e3 - e2), ("certify/rescan/store", en - e3), ("total", en - st)):
    print(f"  {name:22s} min {d.min():6.2f} med {np.median(d):6.2f} p95 {np.percentile(d, 95):6.2f} max {d.max():6.2f} us")
late = np.argsort(en)[-5:]
for w in late:
    print(f"   WG {w}: start {st[w]:.1f} q {e1[w]-st[w]:.1f} s1 {e2[w]-e1[w]:.1f} s2 {e3[w]-e2[w]:.1f} tail {en[w]-e3[w]:.1f} end {en[w]:.1f}")
    if a[w, 6]:
        r = [us(a[w, k]) for k in (6, 7, 8, 9)]
        print(f"      rescan: first item setup done +{r[0]-e3[w]:.1f}, prefilter +{r[1]-r[0]:.1f} (survivors {a[w,10]}), items done +{r[2]-r[1]:.1f}, final evaluate+store +{en[w]-r[2]:.1f}")
