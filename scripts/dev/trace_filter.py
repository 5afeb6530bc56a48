"""Dev diagnostics: per-workgroup start/end/placement of the knn filter kernel."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sfm_mvs_amd import ops, _lib
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
q = torch.rand((nq, 128)).cuda(); t = torch.rand((nt, 128)).cuda()
pm = ops.PairMatcher(nq, nt, q.device)
for _ in range(3): pm.run(q, t)
tr = torch.zeros((8192, 4), dtype=torch.int64, device="cuda")   # 32768 slots: filter 0.., 8192.., 12288.., 14336..; refine 16384..
_lib.lib().sfm_debug_set_trace(ctypes.c_void_p(tr.data_ptr()))
pm.run(q, t); torch.cuda.synchronize()
_lib.lib().sfm_debug_set_trace(None)
G = int(pm.stats[1].item()); a = tr[:G].cpu().numpy()
t0 = a[:, 0].min(); st = (a[:, 0] - t0) / 100.0; en = (a[:, 1] - t0) / 100.0   # us
hw = a[:, 2]; xcc = a[:, 3] & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = xcc * 10000 + se * 100 + sh * 16 + cu
b = tr.view(-1)[8192:8192 + 4 * G].view(G, 4).cpu().numpy()
print("prologue (first segment) us: med %.1f max %.1f | last loop end -> WG end us: med %.1f max %.1f" % (np.median(b[:, 0]) / 100.0, b[:, 0].max() / 100.0, np.median(a[:, 1] - b[:, 1]) / 100.0, (a[:, 1] - b[:, 1]).max() / 100.0))
mhz = (b[:, 3] - b[:, 2]) / ((a[:, 1] - a[:, 0]) / 100.0)
print("shader clock during the kernel (clock64 ticks per us): min %.0f med %.0f max %.0f" % (mhz.min(), np.median(mhz), mhz.max()))
print("G", G, "start us: min %.1f max %.1f | end us: min %.1f max %.1f | dur: min %.1f med %.1f max %.1f" % (st.min(), st.max(), en.min(), en.max(), (en-st).min(), np.median(en-st), (en-st).max()))
u, c = np.unique(key, return_counts=True)
print("distinct CUs used:", len(u), "blocks/CU histogram:", dict(zip(*np.unique(c, return_counts=True))))
print("blocks per XCC:", dict(zip(*np.unique(xcc, return_counts=True))))
late = st > 5
print("late starters:", late.sum(), "their start range", (st[late].min() if late.any() else 0), (st[late].max() if late.any() else 0))
for k in u[c == c.max()][:3]:
    sel = key == k; print(" CU", k, "blocks", np.flatnonzero(sel), "start", st[sel].round(1), "end", en[sel].round(1))

# do the slow workgroups coincide with the ones that cross a query-row-block boundary (second prologue)?
tiles = (nt + 31) // 32; n_rb = (nq + 511) // 512; units = n_rb * tiles
ub = np.array([units * b // G for b in range(G + 1)])
cross = (ub[:-1] // tiles) != ((ub[1:] - 1) // tiles)
d = en - st
print("crossing WGs: %d, their duration med %.1f us; others med %.1f us (max %.1f)" % (cross.sum(), np.median(d[cross]), np.median(d[~cross]), d[~cross].max()))
