"""Dev: frames/s of sift.SiftPipeline on the benchmark frame (968 x 648, seed 3), one frame and three frames in flight, plus the
descriptor kernel alone (the library's event pair) — run once per library (SFM_HIP_LIB) on the SAME box, not under rocprof."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from datagen import scene_image  # noqa: E402
from sfm_mvs_amd import ops, sift  # noqa: E402

dev = torch.device("cuda:0")
w, h = 968, 648
gray = torch.as_tensor(scene_image(w, h, 3)).to(dev)
res = []
for depth in (1, 3):
    pipe = sift.SiftPipeline(w, h, dev, depth=depth)
    for _ in range(6):
        pipe.submit(gray, after=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 60
    for _ in range(n):
        pipe.submit(gray, after=False)
    torch.cuda.synchronize()
    res.append(n / (time.perf_counter() - t0))
eng = pipe.engines[0]
ops.profile_read(7)
ops.profile_enable(True)
for _ in range(10):
    eng.launch(gray)
ms, cnt = ops.profile_read(7)
ops.profile_enable(False)
print("%-40s keypoints %d  frames/s: one stream %6.0f  three in flight %6.0f   descriptor_kernel alone %.1f us" % (
    os.path.basename(os.environ.get("SFM_HIP_LIB", "libsfmhip.so")), int(eng.count[0].item()), res[0], res[1], ms / cnt * 1e3), flush=True)
