"""Dev diagnostics: per-workgroup phase timestamps of refine_q8_body for one BATCHED launch set (SFM_KNN_Q8=1)."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_mvs_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nq = nt = 10000
gen = lambda seed, m: torch.rand((m, 128), generator=torch.Generator().manual_seed(seed)).cuda()
pairs = [(gen(2 * b, nq), gen(2 * b + 1, nt)) for b in range(B)]
bm = ops.BatchMatcher(nq, nt, pairs[0][0].device, batch=B)
for _ in range(5): bm.run(pairs)
nwg = B * ((nq + 15) // 16)
tr = torch.zeros(16384 + 16 * nwg + 64, dtype=torch.int64, device="cuda")
_lib.lib().sfm_debug_set_trace(ctypes.c_void_p(tr.data_ptr()))
bm.run(pairs); torch.cuda.synchronize()
_lib.lib().sfm_debug_set_trace(None)
a = tr[16384:16384 + 16 * nwg].view(nwg, 16).cpu().numpy()
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
st, e1, e2, e3, e4, en = (us(a[:, k]) for k in range(6))
print(f"refine WGs {nwg}: start min {st.min():.1f} med {np.median(st):.1f} max {st.max():.1f} | end min {en.min():.1f} med {np.median(en):.1f} max {en.max():.1f}")
for name, d in (("keys + thresholds", e1 - st), ("listing", e2 - e1), ("integer rows", e3 - e2), ("float32 rows", e4 - e3), ("certify/rescan/store", en - e4), ("total", en - st)):
    print(f"  {name:22s} min {d.min():6.2f} med {np.median(d):6.2f} p95 {np.percentile(d, 95):6.2f} max {d.max():6.2f} us")
print("  records listed (query slot 0 of each WG): mean %.2f p95 %d max %d; rows to float32: mean %.2f max %d" % (a[:, 13].mean(), np.percentile(a[:, 13], 95), a[:, 13].max(), a[:, 14].mean(), a[:, 14].max()))
T = en.max()
for x in np.arange(0, T, max(T / 12, 1e-3)):
    print(f"   t={x:6.1f} us: active WGs {int(((st <= x) & (en > x)).sum()):5d}  started {int((st <= x).sum()):5d} done {int((en <= x).sum()):5d}")
resc = a[:, 6] != 0
print("WGs whose wave 0 rescans:", int(resc.sum()), " their total med %.1f max %.1f" % (np.median((en - st)[resc]) if resc.any() else 0, (en - st)[resc].max() if resc.any() else 0))
print("stats", bm.stats[0].cpu().tolist())
