import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import datagen
from sfm_mvs_amd import pipeline as pl
images, K, P = datagen.layered_views(6, 480, 360, 4)
t = time.time(); feats = pl.features_from_images(images); print("sift", time.time() - t, [len(f[0]) for f in feats])
t = time.time(); out = pl.run_sfm(feats, K, images=images, log=print); print("sfm", time.time() - t)
pose = out["posearr"][9:].reshape(-1, 3, 4)
for k, Pk in enumerate(pose):
    R, tvec = datagen.decompose_P(K, Pk)
    print(k, np.round(R, 3).ravel()[:9:4], np.round(tvec, 3), "center", np.round(-R.T @ tvec, 3))
print("errors", out["errors"], "cloud", out["Xtot"].shape)
Z = out["Xtot"][1:, 2]; print("depth quantiles", np.quantile(Z, [0.05, 0.25, 0.5, 0.75, 0.95]))
