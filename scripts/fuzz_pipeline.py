"""Randomised parity sweep of the WHOLE driver on the GPU box (sfm.py:274-423 through pipeline.run_sfm): the HIP back-end
against the same driver with every operator replaced by the CPU oracle (tests/oracle_backend.py), both running FREE.

  python scripts/fuzz_pipeline.py [seconds] [seed]

One case = one synthetic Gustav-geometry sequence (tests/datagen.gustav_scene: cameras of the reference's pose.csv looking at
its own cloud): a random start camera is not available (the sequence starts at camera 0), so what varies is the LENGTH
(3 .. 16 cameras), the seed of descriptors / clutter, the amount of clutter (0 .. 1 500 features), the descriptor noise
(0 .. 6 grey levels: more ratio-test failures and wrong matches), the pixel noise (0 .. 1.5 px).  Bars (tests/test_gpu_pipeline.py::test_free_running_chain_drift): the same
shapes (= the same integer decisions: match lists, associations, RANSAC inlier sets), poses, cloud and per-frame errors
within 1e-4 relative — measured differences are printed; an exception must be raised by both sides or by neither (a noisy
synthetic sequence can run out of new points, where the reference divides by len(p) = 0).
"""
import faulthandler
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sfm_mvs_amd
from sfm_mvs_amd import _lib, pipeline as pl
from oracle import oracle as O
from datagen import gustav_scene
from oracle_backend import oracle_pipeline_backend

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
O.lib(); sfm_mvs_amd.lib()
be = oracle_pipeline_backend(O)
t0 = time.time(); cases = bad = both_raised = identical = with_ba = 0
worst = dict(P=0.0, error=0.0, cloud=0.0)
while time.time() - t0 < budget:
    n = int(rng.integers(3, 17))
    s = int(rng.integers(1 << 30))
    clutter = int(rng.choice([0, 50, 200, 600, 1500]))
    dn = float(rng.choice([0.0, 1.5, 1.5, 3.0, 6.0]))
    pn = float(rng.choice([0.0, 0.0, 0.2, 0.5, 1.5]))
    ba = False          # (the reference's bundle_adjustment branch is SciPy's dense TRF with finite-difference Jacobians over the whole new cloud:
    # minutes per frame at these sizes; it is covered teacher-forced by tests/test_gpu_pipeline.py::test_driver_with_bundle_adjustment_enabled)
    tag = f"cameras {n} seed {s} clutter {clutter} desc_noise {dn} pix_noise {pn} ba {ba}"
    tc = time.time()
    print("case", tag, flush=True) if os.environ.get("FZ_VERBOSE") else None
    faulthandler.cancel_dump_traceback_later(); faulthandler.dump_traceback_later(150, exit=True)   # a stalled case: every thread's stack, then exit
    K, P, feats, ids = gustav_scene(n, seed=s, clutter=clutter, desc_noise=dn, pix_noise=pn)
    got = want = eg = ew = None
    try:
        got = pl.run_sfm(feats, K, bundle_adjustment=ba)
    except Exception as e:  # noqa: BLE001
        eg = e
    try:
        want = pl.run_sfm(feats, K, be=be, bundle_adjustment=ba)
    except Exception as e:  # noqa: BLE001
        ew = e
    cases += 1; with_ba += ba
    if time.time() - tc > 30: print(f"(slow case, {time.time() - tc:.0f} s: {tag})", flush=True)
    msg = None
    if (eg is None) != (ew is None):
        msg = f"raised on one side only: hip {eg!r} / oracle {ew!r}"
    elif eg is not None:
        both_raised += 1
        if type(eg) is not type(ew):
            msg = f"different exceptions: hip {eg!r} / oracle {ew!r}"
    else:
        if not (got["posearr"].shape == want["posearr"].shape and got["Xtot"].shape == want["Xtot"].shape and len(got["errors"]) == len(want["errors"])):
            msg = f"shapes differ: cloud {got['Xtot'].shape} vs {want['Xtot'].shape}, errors {len(got['errors'])} vs {len(want['errors'])}"
        else:
            m = len(want["posearr"][9:]) // 12
            dP = (np.abs(got["posearr"] - want["posearr"])[9:].reshape(m, 12).max(1) / np.abs(want["posearr"][9:]).reshape(m, 12).max(1)).max()
            dE = max([abs(a - b) / max(abs(b), 1e-300) for a, b in zip(got["errors"], want["errors"])] + [0.0])
            dX = np.abs(got["Xtot"] - want["Xtot"]).max() / max(np.abs(want["Xtot"]).max(), 1e-300) if len(want["Xtot"]) else 0.0
            worst["P"] = max(worst["P"], float(dP)); worst["error"] = max(worst["error"], float(dE)); worst["cloud"] = max(worst["cloud"], float(dX))
            identical += bool(np.array_equal(got["posearr"], want["posearr"]) and np.array_equal(got["Xtot"], want["Xtot"]))
            if not (dP <= 1e-4 and dE <= 1e-4 and dX <= 1e-4):
                msg = f"differences beyond 1e-4: P {dP:.3g} error {dE:.3g} cloud {dX:.3g}"
    if msg:
        bad += 1
        print("MISMATCH", tag + ":", msg[:400], flush=True)
faulthandler.cancel_dump_traceback_later()
print(f"fuzz_pipeline: seed {seed}, {cases} sequences ({with_ba} with bundle adjustment, {both_raised} raised on both sides, {identical} bit-identical poses + cloud), "
      f"worst relative differences {worst}, {bad} mismatches, {time.time() - t0:.0f} s; build {_lib.build_id()}")
