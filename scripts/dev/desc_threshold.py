"""Dev: frames/s of sift.SiftPipeline (3 frames in flight, and one) on procedural frames of several sizes — run it with a DEV
library under SFM_SIFT_DESC_ROWS_MIN=0 (row-group body always) and =1000000 (cell body always) to place kDescRowsMin.

usage: SFM_HIP_LIB=.../libsfmhip_dev.so SFM_SIFT_DESC_ROWS_MIN=0 python scripts/dev/desc_threshold.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from datagen import scene_image  # noqa: E402
from sfm_mvs_amd import sift  # noqa: E402

dev = torch.device("cuda:0")
for (w, h) in ((400, 268), (560, 376), (684, 458), (800, 536), (968, 648), (1368, 916)):
    gray = torch.as_tensor(scene_image(w, h, 3)).to(dev)
    res = []
    for depth in (1, 3):
        pipe = sift.SiftPipeline(w, h, dev, depth=depth)
        for _ in range(6):
            pipe.submit(gray, after=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        for _ in range(n):
            pipe.submit(gray, after=False)
        torch.cuda.synchronize()
        res.append(n / (time.perf_counter() - t0))
        nkp = int(pipe.engines[0].count[0].item())
    print("%4d x %4d  keypoints %6d  frames/s depth1 %7.0f  depth3 %7.0f   rows_min=%s" % (w, h, nkp, res[0], res[1], os.environ.get("SFM_SIFT_DESC_ROWS_MIN")), flush=True)
