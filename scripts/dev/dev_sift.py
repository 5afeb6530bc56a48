"""Dev check: HIP SIFT vs the CPU oracle on procedural images (run on the GPU box)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from sfm_mvs_amd import sift
from oracle import oracle as orc
import datagen

for (w, h, seed) in [(160, 120, 0), (256, 192, 1), (333, 217, 2), (968, 648, 3)]:
    g = datagen.scene_image(w, h, seed)
    t = time.time(); kpo, deso = orc.sift(g); tc = time.time() - t
    eng = sift.Sift(w, h, "cuda")
    gd = torch.as_tensor(g).cuda()
    kp, des = eng.run(gd); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5): eng.launch(gd)
    torch.cuda.synchronize(); tg = (time.time() - t) / 5
    kp, des = kp.cpu().numpy(), des.cpu().numpy()
    print(f"{w}x{h}: oracle {len(kpo)} kp in {tc:.2f}s | hip {len(kp)} kp in {tg*1e3:.2f} ms | counts {eng.count.tolist()}")
    if len(kp) == len(kpo):
        same_kp = (kp.view(np.int32) == kpo.view(np.int32)).all(axis=1)
        same_des = (des == deso).all(axis=1)
        print("   identical keypoints:", same_kp.sum(), "identical descriptors:", same_des.sum(), "max |ddesc|", np.abs(des - deso).max())
        bad = np.flatnonzero(~same_kp)[:5]
        for b in bad: print("   kp", b, kp[b, :5], kpo[b, :5])
    else:
        a = {tuple(r) for r in kp[:, :4].view(np.int32).tolist()}; b = {tuple(r) for r in kpo[:, :4].view(np.int32).tolist()}
        print("   common", len(a & b), "only hip", len(a - b), "only oracle", len(b - a))
