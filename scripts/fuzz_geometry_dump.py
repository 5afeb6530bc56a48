"""scripts/fuzz_geometry.py with the inputs of every solvePnPRansac case whose result it reports as a MISMATCH dumped to
gpurun_out/geometry_case_<k>.npz (objectPoints, imagePoints, K, parameters, both sides' results) — a seed then gives files to study offline."""
import os, sys, runpy
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sfm_mvs_amd import ransac
from oracle import oracle as O
_orig_h, _orig_o = ransac.solve_pnp_ransac, O.solve_pnp_ransac
last = {}


def hip(*a, **k):
    r = _orig_h(*a, **k)
    last["hip"] = (a, k, r)
    return r


def orc(*a, **k):
    r = _orig_o(*a, **k)
    last["orc"] = (a, k, r)
    return r


ransac.solve_pnp_ransac, O.solve_pnp_ransac = hip, orc
import builtins
_print, n_dump = builtins.print, [0]


def spy(*a, **k):
    _print(*a, **k)
    line = " ".join(str(x) for x in a)
    if line.startswith("MISMATCH ransac") and "PnP pose differs" in line and "hip" in last and "orc" in last:
        (ah, kh, rh), (ao, ko, ro) = last["hip"], last["orc"]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", f"geometry_case_{n_dump[0]}.npz"), X=np.asarray(ah[0]), uv=np.asarray(ah[1]), K=np.asarray(ah[2]),
                 its=kh.get("iterations_count"), rep=kh.get("reprojection_error"), conf=kh.get("confidence"),
                 r_h=np.asarray(rh[1]), t_h=np.asarray(rh[2]), inl_h=np.asarray(rh[3]), info_h=np.asarray(rh[4]),
                 r_o=np.asarray(ro[1]), t_o=np.asarray(ro[2]), inl_o=np.asarray(ro[3]), model_o=np.asarray(ro[4]), st_o=ro[5], tag=line)
        n_dump[0] += 1


builtins.print = spy
runpy.run_path(os.path.join(ROOT, "scripts", "fuzz_geometry.py"), run_name="__main__")
