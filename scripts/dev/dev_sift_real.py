"""Dev: SIFT timing on a frame with a photograph-like keypoint count (~2-3k at 968x648)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sfm_mvs_amd import sift
import datagen
from scipy.ndimage import gaussian_filter
g = datagen.scene_image(968, 648, 3)
g = np.clip(np.rint(gaussian_filter(g.astype(np.float64), float(sys.argv[1]) if len(sys.argv) > 1 else 2.0)), 0, 255).astype(np.uint8)
d = torch.as_tensor(g).cuda()
eng = sift.Sift(968, 648, "cuda")
for _ in range(3): eng.launch(d)
torch.cuda.synchronize(); t = time.time()
for _ in range(20): eng.launch(d)
torch.cuda.synchronize(); print("single stream ms/frame", (time.time() - t) / 20 * 1e3, eng.count.tolist())
for depth in (2, 3, 4, 6, 8):
    pipe = sift.SiftPipeline(968, 648, "cuda", depth=depth)
    for _ in range(2 * depth): pipe.submit(d, after=False)
    pipe.synchronize(); t = time.time()
    for _ in range(120): pipe.submit(d, after=False)
    pipe.synchronize(); print("depth", depth, "ms/frame", (time.time() - t) / 120 * 1e3)
