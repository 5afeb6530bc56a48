"""RANSAC entry points of the path — cv2.findEssentialMat (sfm.py:307), cv2.recoverPose (sfm.py:311),
cv2.solvePnPRansac (sfm.py:67) — as thin wrappers over ONE library call each (`sfm_find_essential_mat`,
`sfm_recover_pose`, `sfm_solve_pnp_ransac`, csrc/ransac.hip).

The library generates hypotheses on the host in chunks (OpenCV's RNG and subsets, five-point / EPnP on five
correspondences each), scores a whole chunk against every correspondence in one launch, replays OpenCV's sequential
bookkeeping over the counts and runs the Levenberg-Marquardt sweeps of the PnP refinement on the device; masks and
inlier lists stay in HBM until the caller asks for them.  Points may be NumPy arrays (uploaded once) or CUDA tensors.
There is no CPU path: the functions raise without a HIP device.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import SfmHipError, check, on_device, ptr, stream_ptr

_vp = ctypes.c_void_p
_ws = {}


def _dev():
    if not torch.cuda.is_available():
        raise SfmHipError("sfm_mvs_amd.ransac needs a HIP device: there is no CPU fallback")
    return torch.device("cuda")


def _workspace(device, nbytes):
    """Grow-only scratch per (device, stream): the entry points synchronise their stream before returning, so a buffer
    is never in use by two calls of the same stream."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, _lib.raw_stream(idx))
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _ws[key] = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
    return buf


def _points(a, cols):
    """(n, cols) float32 on the device, contiguous; NumPy input of any cv2-accepted shape ((n,1,cols), (n,cols))."""
    if torch.is_tensor(a):
        if not a.is_cuda:
            raise SfmHipError("tensor inputs must be CUDA tensors")
        return a.reshape(-1, cols).to(torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1, cols))).to(_dev())


def _k(K):
    return np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))


def find_essential_mat(pts0, pts1, K, prob=0.999, threshold=1.0, max_iters=1000, return_device_mask=False, want_info=False):
    """cv2.findEssentialMat(points1, points2, K, method=RANSAC, prob, threshold) -> (E (3,3) float64, mask (N,1) uint8 in
    {0,1}); (None, None) when no model was found.  With exactly five points OpenCV returns all the solver's models
    stacked ((3k,3)) and an all-ones mask — so does this."""
    p0, p1 = _points(pts0, 2), _points(pts1, 2)
    n = p0.shape[0]
    if p1.shape[0] != n:
        raise SfmHipError("findEssentialMat: the two point sets differ in length")
    Kc = _k(K)
    E, info = np.zeros(90), np.zeros(4, np.int32)
    lib = _lib.lib()
    dev = p0.device
    mask = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
    ws = _workspace(dev, lib.sfm_find_essential_mat_ws_bytes(n))
    with on_device(dev):
        check(lib.sfm_find_essential_mat(ptr(p0), ptr(p1), n, Kc.ctypes.data_as(_vp), float(prob), float(threshold), int(max_iters),
                                         E.ctypes.data_as(_vp), info.ctypes.data_as(_vp), ptr(mask), ptr(ws), ws.numel(),
                                         stream_ptr()), "sfm_find_essential_mat")
    k = int(info[0])
    if k <= 0:
        return (None, None, info) if want_info else (None, None)
    m = mask[:n].reshape(-1, 1)
    out = (E[:9 * k].reshape(3 * k, 3).copy(), m if return_device_mask else m.cpu().numpy())
    return out + (info,) if want_info else out


def recover_pose(E, pts0, pts1, K, distance_thresh=50.0, rows=4, return_device_mask=False):
    """cv2.recoverPose(E, points1, points2, K) -> (good, R (3,3), t (3,1), mask (N,1) uint8 in {0,255})."""
    p0, p1 = _points(pts0, 2), _points(pts1, 2)
    n = p0.shape[0]
    Ec = np.ascontiguousarray(np.asarray(E, np.float64).reshape(-1)[:9])
    Kc = _k(K)
    R, t, good = np.empty(9), np.empty(3), np.zeros(1, np.int32)
    lib = _lib.lib()
    dev = p0.device
    mask = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
    ws = _workspace(dev, lib.sfm_recover_pose_ws_bytes(n))
    with on_device(dev):
        check(lib.sfm_recover_pose(Ec.ctypes.data_as(_vp), ptr(p0), ptr(p1), n, Kc.ctypes.data_as(_vp), float(distance_thresh), int(rows),
                                   R.ctypes.data_as(_vp), t.ctypes.data_as(_vp), good.ctypes.data_as(_vp), ptr(mask), ptr(ws),
                                   ws.numel(), stream_ptr()), "sfm_recover_pose")
    m = mask[:n].reshape(-1, 1)
    return int(good[0]), R.reshape(3, 3), t.reshape(3, 1), (m if return_device_mask else m.cpu().numpy())


def solve_pnp_ransac(X, uv, K, iterations_count=100, reprojection_error=8.0, confidence=0.99, return_device_inliers=False,
                     want_info=False):
    """cv2.solvePnPRansac(objectPoints, imagePoints, K, dist) with all defaults (the reference's 5th positional argument
    lands in `rvec` and is ignored: SURVEY 3.6-1): RANSAC over EPnP on 5-point samples, then solvePnP(ITERATIVE) on the
    inlier set (exactly 4 points: solvePnP(P3P); exactly 5: solvePnP(EPNP)).  Returns (ok, rvec (3,1), tvec (3,1), inliers (k,1) int32 or None)."""
    Xd, ud = _points(X, 3), _points(uv, 2)
    n = Xd.shape[0]
    if ud.shape[0] != n:
        raise SfmHipError("solvePnPRansac: object and image points differ in length")
    if n < 4:          # OpenCV asserts npoints >= 4 (exactly 4: solvePnP(P3P), every point an inlier)
        raise SfmHipError(f"solvePnPRansac: at least 4 correspondences are required (got {n})")
    Kc = _k(K)
    r, t, info = np.zeros(3), np.zeros(3), np.zeros(4, np.int32)
    lib = _lib.lib()
    dev = Xd.device
    inl = torch.empty(n, dtype=torch.int32, device=dev)
    ws = _workspace(dev, lib.sfm_solve_pnp_ransac_ws_bytes(n))
    with on_device(dev):
        check(lib.sfm_solve_pnp_ransac(ptr(Xd), ptr(ud), n, Kc.ctypes.data_as(_vp), int(iterations_count), float(reprojection_error),
                                       float(confidence), r.ctypes.data_as(_vp), t.ctypes.data_as(_vp), info.ctypes.data_as(_vp),
                                       ptr(inl), ptr(ws), ws.numel(), stream_ptr()), "sfm_solve_pnp_ransac")
    if not info[0]:
        return (False, None, None, None, info) if want_info else (False, None, None, None)
    il = inl[:int(info[1])].reshape(-1, 1)
    out = (True, r.reshape(3, 1), t.reshape(3, 1), il if return_device_inliers else il.cpu().numpy())
    return out + (info,) if want_info else out
