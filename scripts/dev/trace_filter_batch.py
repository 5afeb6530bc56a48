"""Dev diagnostics: per-workgroup start/end/placement of the knn filter kernel for one BATCHED launch: how evenly do the
XCDs finish (equal work per workgroup, unequal clocks)?"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sfm_mvs_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nq = nt = 10000
q = torch.rand((nq, 128), generator=torch.Generator().manual_seed(0)).cuda()
t = torch.rand((nt, 128), generator=torch.Generator().manual_seed(1)).cuda()
bm = ops.BatchMatcher(nq, nt, q.device, batch=B)
for _ in range(20): bm.run([(q, t)] * B)
tr = torch.zeros(16384 + 16 * B * 625 + 64, dtype=torch.int64, device="cuda")
for rep in range(3):
    tr.zero_()
    _lib.lib().sfm_debug_set_trace(ctypes.c_void_p(tr.data_ptr()))
    bm.run([(q, t)] * B); torch.cuda.synchronize()
    _lib.lib().sfm_debug_set_trace(None)
    G = int(bm.stats[0, 1].item())
    a = tr[:4 * G].view(G, 4).cpu().numpy()
    b = tr[8192:8192 + 4 * G].view(G, 4).cpu().numpy()
    t0 = a[:, 0].min(); st = (a[:, 0] - t0) / 100.0; en = (a[:, 1] - t0) / 100.0
    xcc = a[:, 3] & 0xF
    mhz = (b[:, 3] - b[:, 2]) / np.maximum((a[:, 1] - a[:, 0]) / 100.0, 1e-3)
    print(f"rep {rep}: G {G}  start max {st.max():.1f} us | end min {en.min():.1f} med {np.median(en):.1f} max {en.max():.1f} us | duration med {np.median(en-st):.1f}")
    for x in range(8):
        s = xcc == x
        if s.any():
            print(f"   XCD {x}: {int(s.sum()):3d} WGs  end med {np.median(en[s]):6.1f} max {en[s].max():6.1f} us   clock med {np.median(mhz[s]):5.0f} MHz")
    print(f"   idle tail: sum over WGs of (kernel end - WG end) = {((en.max() - en).sum() / G):.1f} us average per WG = {100 * (en.max() - en).mean() / en.max():.1f} % of the launch")
    pro = b[:, 0] / 100.0                                   # (accumulated: start -> loop begin, once per segment)
    loop_done = (b[:, 1] - t0) / 100.0
    dur = en - st
    print(f"   per WG: duration med {np.median(dur):.1f} min {dur.min():.1f} max {dur.max():.1f} us | start->first loop (+ later segments, accumulated) med {np.median(pro):.1f} us | "
          f"last loop end -> WG end med {np.median(en - loop_done):.2f} us | kernel = {en.max():.1f} us, first WG start spread {st.max():.1f} us")
    cyc = (b[:, 3] - b[:, 2])
    print(f"   shader cycles per WG med {np.median(cyc):.0f}  (x{G} WGs / 100160 unit tile-steps = {np.median(cyc) * G / (B * 40 * 313):.0f} cycles per tile-step incl. everything)")
    print("   WG start percentiles (us): " + " ".join(f"p{p}={np.percentile(st, p):.1f}" for p in (10, 25, 50, 75, 90, 99, 100)) +
          " | by XCD median: " + " ".join(f"{np.median(st[xcc == x]):.1f}" for x in range(8)))
