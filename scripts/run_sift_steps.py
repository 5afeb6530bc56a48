"""Dev driver: N SIFT detectAndCompute launches on one procedural frame (for rocprofv3 / timing)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from sfm_mvs_amd import sift
import datagen
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w = int(sys.argv[2]) if len(sys.argv) > 2 else 968
h = int(sys.argv[3]) if len(sys.argv) > 3 else 648
img = datagen.scene_image(w, h, 3)
if os.environ.get("SFM_SIFT_SMOOTH"):      # photograph-like keypoint count (~1-3k instead of 15k)
    import numpy as np
    from scipy.ndimage import gaussian_filter
    img = np.clip(np.rint(gaussian_filter(img.astype(np.float64), float(os.environ["SFM_SIFT_SMOOTH"]))), 0, 255).astype(np.uint8)
g = torch.as_tensor(img).cuda()
eng = sift.Sift(w, h, "cuda")
for _ in range(3): eng.launch(g)
torch.cuda.synchronize(); t = time.time()
for _ in range(steps): eng.launch(g)
torch.cuda.synchronize(); dt = (time.time() - t) / steps
print(f"done {w}x{h}: {dt*1e3:.3f} ms/frame, counts {eng.count.tolist()}")
