"""Dev: time the guarded triangulation (pass 1 + fixup) at 1e7 points for the library named by SFM_HIP_LIB, check it against
the faithful kernel on the first 2e6 points."""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from datagen import load_pose_csv
from sfm_mvs_amd import ops
K, P = load_pose_csv()
n = 10_000_000
g = torch.Generator(device="cuda").manual_seed(2)
X = torch.stack([torch.rand(n, generator=g, device="cuda") * 9.9 - 6.3, torch.rand(n, generator=g, device="cuda") * 7.6 - 2.6,
                 torch.rand(n, generator=g, device="cuda") * 9.8 + 3.2, torch.ones(n, device="cuda")]).double()
xs = []
for Pm in (P[1], P[2]):
    x = torch.from_numpy(Pm).cuda() @ X
    xs.append(((x[:2] / x[2]) + 0.3 * torch.randn((2, n), generator=g, device="cuda", dtype=torch.float64)).float().contiguous())
a, b = xs
for _ in range(3): ops.triangulate(P[1], P[2], a, b, normalise_w="guarded")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): out = ops.triangulate(P[1], P[2], a, b, normalise_w="guarded")
e1.record(); torch.cuda.synchronize()
m = 2_000_000
ref = ops.triangulate(P[1], P[2], a[:, :m], b[:, :m], normalise_w=True)
got = ops.triangulate(P[1], P[2], a[:, :m], b[:, :m], normalise_w="guarded")
print(os.environ.get("SFM_HIP_LIB", "default"), "guarded 1e7: %.1f us" % (e0.elapsed_time(e1) / 10 * 1e3), "bit-identical on 2e6:", bool(torch.equal(ref.view(torch.int32), got.view(torch.int32))))
