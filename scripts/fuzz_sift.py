"""Randomised SIFT parity sweep on the GPU box: HIP path vs the CPU oracle, bit for bit, over random frame sizes,
contents and detector parameters.  Usage: python scripts/fuzz_sift.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sfm_mvs_amd import sift
from oracle import oracle as orc
import datagen

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
t0 = time.time(); cases = bad = kps = 0
while time.time() - t0 < budget:
    w, h = int(rng.integers(24, 520)), int(rng.integers(24, 420))
    kind = int(rng.integers(0, 5))
    s = int(rng.integers(0, 1 << 30))
    if kind == 0:   g = datagen.scene_image(w, h, s)
    elif kind == 1: g = rng.integers(0, 256, (h, w), dtype=np.uint8)                                   # white noise
    elif kind == 2: g = np.kron(rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8), dtype=np.uint8), np.ones((8, 8), np.uint8))[:h, :w].copy()  # blocks: exact ties
    elif kind == 3: g = np.rot90(datagen.scene_image(h, w, s)).copy()
    else:           g = (datagen.scene_image(w, h, s) // 16 * 16).astype(np.uint8)                    # posterised: plateaus
    nl = int(rng.choice([3, 3, 3, 2, 4, 5]))
    ct = float(rng.choice([0.04, 0.04, 0.02, 0.08]))
    et = float(rng.choice([10.0, 10.0, 5.0, 20.0]))
    sg = float(rng.choice([1.6, 1.6, 1.2, 2.0]))
    kpo, deso = orc.sift(g, n_octave_layers=nl, contrast=ct, edge=et, sigma=sg)
    try:
        eng = sift.Sift(w, h, "cuda", n_octave_layers=nl, contrast_threshold=ct, edge_threshold=et, sigma=sg, max_keypoints=1 << 16)
        kp, des = eng.run(torch.as_tensor(g).cuda())
        kp, des = kp.cpu().numpy(), des.cpu().numpy()
        ok = kp.shape == kpo.shape and np.array_equal(kp.view(np.int32), kpo.view(np.int32)) and np.array_equal(des, deso)
    except Exception as e:                                                                            # noqa: BLE001
        ok = False; print("EXC", repr(e)[:200])
    cases += 1; kps += len(kpo)
    if not ok:
        bad += 1
        print(f"MISMATCH case {cases}: {w}x{h} kind {kind} seed {s} nL {nl} ct {ct} et {et} sigma {sg}: oracle {len(kpo)} hip {len(kp) if 'kp' in dir() else '?'}")
from sfm_mvs_amd import _lib as _sfm_lib
print(f"fuzz_sift: {cases} cases, {kps} keypoints compared, {bad} mismatches, {time.time() - t0:.0f} s; build {_sfm_lib.build_id()}")
