"""Dev: one gustav_scene sequence through pipeline.run_sfm on the HIP back-end, with a watchdog that dumps the Python stack if it stalls.
usage: python scripts/dev/fz_case.py n seed clutter desc_noise pix_noise [ba]"""
import faulthandler
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sfm_mvs_amd import pipeline as pl
from datagen import gustav_scene

n, s, clutter = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dn, pn = float(sys.argv[4]), float(sys.argv[5])
ba = len(sys.argv) > 6 and sys.argv[6] == "1"
K, P, feats, ids = gustav_scene(n, seed=s, clutter=clutter, desc_noise=dn, pix_noise=pn)
faulthandler.dump_traceback_later(45, exit=True)
t = time.time()
try:
    out = pl.run_sfm(feats, K, bundle_adjustment=ba, log=print if os.environ.get("FZ_LOG") else None)
    print("ok", time.time() - t, out["Xtot"].shape, flush=True)
except Exception as e:  # noqa: BLE001
    print("exc", repr(e)[:300], time.time() - t, flush=True)
