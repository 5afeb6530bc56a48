"""Dev diagnostics: per-workgroup phase timestamps of knn_refine_kernel for one BATCHED launch set."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sfm_mvs_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nq = nt = 10000
q = torch.rand((nq, 128), generator=torch.Generator().manual_seed(0)).cuda()
t = torch.rand((nt, 128), generator=torch.Generator().manual_seed(1)).cuda()
bm = ops.BatchMatcher(nq, nt, q.device, batch=B)
for _ in range(3): bm.run([(q, t)] * B)
nwg = B * ((nq + 15) // 16)
tr = torch.zeros(16384 + 16 * nwg + 64, dtype=torch.int64, device="cuda")
_lib.lib().sfm_debug_set_trace(ctypes.c_void_p(tr.data_ptr()))
bm.run([(q, t)] * B); torch.cuda.synchronize()
_lib.lib().sfm_debug_set_trace(None)
a = tr[16384:16384 + 16 * nwg].view(nwg, 16).cpu().numpy()
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
st, e1, e2, e3, en = (us(a[:, k]) for k in range(5))
print(f"refine WGs {nwg}: start min {st.min():.1f} med {np.median(st):.1f} max {st.max():.1f} | end min {en.min():.1f} med {np.median(en):.1f} max {en.max():.1f}")
for name, d in (("Q rows->LDS", e1 - st), ("sweep1", e2 - e1), ("sweep2", e3 - e2), ("certify/rescan/store", en - e3), ("total", en - st)):
    print(f"  {name:22s} min {d.min():6.2f} med {np.median(d):6.2f} p95 {np.percentile(d, 95):6.2f} max {d.max():6.2f} us")
c0, c1, c2 = us(a[:, 11]), us(a[:, 12]), us(a[:, 15])
for name, d in (("  s2: compaction", c0 - e2), ("  s2: fp16 screen", c1 - c0), ("  s2: exact evaluate", c2 - c1), ("  s2: reduce", e3 - c2)):
    print(f"  {name:22s} min {d.min():6.2f} med {np.median(d):6.2f} p95 {np.percentile(d, 95):6.2f} max {d.max():6.2f} us")
print("  rows listed (query slot 0 of each WG): mean %.1f p95 %d max %d; after the screen: mean %.2f max %d" % (a[:, 13].mean(), np.percentile(a[:, 13], 95), a[:, 13].max(), a[:, 14].mean(), a[:, 14].max()))
# concurrency profile
T = en.max()
for x in np.arange(0, T, max(T / 16, 1e-3)):
    print(f"   t={x:6.1f} us: active WGs {int(((st <= x) & (en > x)).sum()):5d}  started {int((st <= x).sum()):5d} done {int((en <= x).sum()):5d}")
hw = a[:, 5]
xcc = (hw >> 0) & 0xF
print("HW_ID samples:", [hex(int(v)) for v in hw[:4]])
resc = a[:, 6] != 0
print("rescanning WGs:", int(resc.sum()), " their total med %.1f max %.1f" % (np.median((en - st)[resc]) if resc.any() else 0, (en - st)[resc].max() if resc.any() else 0))
