"""Generates the golden vectors under tests/golden/ by EXECUTING the reference's own pure-NumPy
functions (AST-extracted from /root/reference/sfm.py — the script itself runs on import and needs
cv2/open3d/GUI, so it cannot be imported) and by copying the reference's DATA artefacts.

Run in the build container only (the reference is not present on the GPU box):
    python tests/golden/make_golden.py
Outputs are data (inputs + expected outputs); no reference source text is stored.
"""
import ast
import contextlib
import io
import os
import shutil
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def extract(names):
    tree = ast.parse(open(os.path.join(REF, "sfm.py")).read())
    ns = {"np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), "sfm.py", "exec"), ns)
    return ns


def helper_goldens():
    """The reference's OWN helper functions (Triangulation sfm.py:45, PnP :60, ReprojectionError :79), AST-extracted and
    executed with `cv2` bound to the CPU oracle's cv2-named facade: pins the helpers' data flow (transposed views,
    homogeneous division, (N,1,3) layouts, the stray positional argument, inlier gathers) — everything of the
    reference that is not inside OpenCV.  Inputs + outputs are stored; the GPU tests replay them on the HIP path."""
    import sys
    root = os.path.dirname(os.path.dirname(OUT))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    from oracle import oracle as O
    from oracle_backend import OracleCv2
    from datagen import decompose_P, gustav_pair
    ns = extract({"Triangulation", "PnP", "ReprojectionError"})
    ns["cv2"] = OracleCv2(O)
    out = {}
    for tag, k, n, sigma in (("a", 1, 400, 0.3), ("b", 30, 250, 1.0)):
        K, P1, P2, X, x1, x2 = gustav_pair(k, n, sigma, seed=100 + k)
        x2[::23] += np.float32(25.0)                                # a few gross outliers for the PnP inlier set
        # sfm.py:313-317: first call with (M,2) arrays, repeat=False
        pts1, pts2, cloud = ns["Triangulation"](P1, P2, x1, x2, K, False)
        R, t = decompose_P(K, P2)
        Rt = np.hstack([R, t.reshape(3, 1)])
        err1, X3, proj1 = ns["ReprojectionError"](cloud, pts2, Rt, K, 1)
        # sfm.py:325: PnP on the bootstrap cloud, initial = 1
        Rp, tp, p_in, X_in, p0_in = ns["PnP"](X3, pts2, K, np.zeros((5, 1), np.float32), pts1, 1)
        # sfm.py:362-366: later frames, initial = 0, then the homogenity = 0 error
        Rq, tq, q_in, Xq_in, q0_in = ns["PnP"](X3[:, 0, :], x2, K, np.zeros((5, 1), np.float32), x1, 0)
        err0, X0, proj0 = ns["ReprojectionError"](Xq_in, q_in, np.hstack([Rq, tq]), K, 0)
        for name, v in dict(K=K, P1=P1, P2=P2, x1=x1, x2=x2, Rt=Rt, pts1=np.ascontiguousarray(pts1), pts2=np.ascontiguousarray(pts2),
                            cloud=cloud, err1=err1, X3=X3, proj1=proj1, Rp=Rp, tp=tp, p_in=p_in, X_in=X_in, p0_in=p0_in,
                            Rq=Rq, tq=tq, q_in=q_in, Xq_in=Xq_in, q0_in=q0_in, err0=err0, proj0=proj0).items():
            out[f"{tag}_{name}"] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "helpers.npz"), **out)


def photo_fixture():
    """The reference's one sample photograph (image.jpg, 1936 x 1296, NIKON D60) as DATA: decoded with PIL, converted to
    grey with the oracle's restatement of cv2.cvtColor(BGR2GRAY), stored as a uint8 array.  The SIFT parity tests run on
    it at full size and at the reference's working size (one pyrDown, sfm.py:40)."""
    import sys
    root = os.path.dirname(os.path.dirname(OUT))
    sys.path.insert(0, root)
    from PIL import Image
    from oracle import oracle as O
    rgb = np.asarray(Image.open(os.path.join(REF, "image.jpg")).convert("RGB"))
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])
    gray = O.bgr2gray(bgr)
    assert gray.shape == (1296, 1936) and gray.dtype == np.uint8
    np.savez_compressed(os.path.join(OUT, "photo_gray.npz"), gray=gray)


def main():
    helper_goldens()
    photo_fixture()
    ns = extract({"common_points", "to_ply"})
    rng = np.random.default_rng(20260928)

    # ---- common_points (sfm.py:215-239): several cases incl. the x-OR-y quirk and duplicates
    cases = {}
    for name, n1, n2, frac in [("a", 60, 80, 0.5), ("b", 200, 150, 0.3), ("c", 5, 7, 0.0), ("d", 40, 40, 1.0)]:
        pts2 = np.round(rng.uniform(0, 900, (n2, 2)), 2).astype(np.float32)
        pts3 = np.round(rng.uniform(0, 900, (n2, 2)), 2).astype(np.float32)
        pts1 = np.round(rng.uniform(0, 900, (n1, 2)), 2).astype(np.float32)
        k = int(frac * min(n1, n2))
        sel1 = rng.permutation(n1)[:k]
        sel2 = rng.permutation(n2)[:k]
        pts1[sel1] = pts2[sel2]
        if name == "a":   # rows that agree in x only / y only, and a duplicated row in pts2
            pts1[0, 0] = pts2[3, 0]
            pts1[1, 1] = pts2[5, 1]
            pts2[9] = pts2[4]
        with contextlib.redirect_stdout(io.StringIO()):
            i1, i2, t1, t2 = ns["common_points"](pts1, pts2, pts3)
        cases[f"{name}_pts1"], cases[f"{name}_pts2"], cases[f"{name}_pts3"] = pts1, pts2, pts3
        cases[f"{name}_indx1"], cases[f"{name}_indx2"] = np.asarray(i1, np.int64), np.asarray(i2, np.int64)
        cases[f"{name}_temp1"], cases[f"{name}_temp2"] = np.asarray(t1), np.asarray(t2)
    np.savez_compressed(os.path.join(OUT, "common_points.npz"), **cases)

    # ---- to_ply (sfm.py:169-201): cloud with far outliers + the leading zero row (quirk 8)
    pts = np.vstack([np.zeros((1, 3)), rng.normal(0, 1.5, (300, 3)) + [0, 1, 8], rng.normal(0, 40, (6, 3))])
    cols = np.vstack([np.zeros((1, 3)), rng.integers(0, 256, (306, 3))]).astype(np.float64)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "Point_Cloud"))
    with contextlib.redirect_stdout(io.StringIO()):
        ns["to_ply"](tmp, pts, cols, False)
    text = open(os.path.join(tmp, "Point_Cloud", "sparse.ply")).read()
    shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(OUT, "to_ply.npz"), points=pts, colors=cols, ply_text=np.array(text))

    # ---- the reference's data artefacts (outputs of its Gustav II Adolf run)
    shutil.copy(os.path.join(REF, "pose.csv"), os.path.join(OUT, "pose.csv"))
    verts = []
    with open(os.path.join(REF, "Point_Cloud", "sparse.ply")) as f:
        lines = f.read().split("\n")
    start = next(i for i, l in enumerate(lines) if "end_header" in l) + 1
    for l in lines[start:]:
        p = l.split()
        if len(p) == 6:
            verts.append([float(v) for v in p])
    verts = np.array(verts)
    assert len(verts) == 19282, len(verts)
    sub = verts[rng.permutation(len(verts))[:4000]]
    np.savez_compressed(os.path.join(OUT, "sparse_ply_subset.npz"), verts=sub, total=np.array(len(verts)))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
