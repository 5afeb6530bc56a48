"""Small host-side geometry helpers of the facade: cv2.Rodrigues (sfm.py:69,84,119), computed by the library's host code
(`sfm_host_rodrigues`, csrc/ransac.hip + host_solvers.h) — the same routines the RANSAC entry points use for their models.

The minimal solvers themselves (five-point, EPnP, the ITERATIVE DLT initialisation) live in the library
(csrc/host_solvers.h) and are reached through `sfm_mvs_amd.ransac`; this module holds no solver code and nothing from oracle/.
"""
import ctypes

import numpy as np

from . import _lib

_vp = ctypes.c_void_p


def rodrigues_vec2mat(rvec, want_jac=False):
    """Rotation vector -> 3x3 matrix (and dR/dr, 3 x 9, when asked)."""
    r = np.ascontiguousarray(np.asarray(rvec, np.float64).reshape(3))
    R, J = np.empty((3, 3)), np.empty((3, 9))
    _lib.check(_lib.lib().sfm_host_rodrigues(r.ctypes.data_as(_vp), 0, R.ctypes.data_as(_vp), J.ctypes.data_as(_vp)), "sfm_host_rodrigues")
    return (R, J) if want_jac else R


def rodrigues_mat2vec(R):
    """3x3 matrix (orthonormalised by SVD first, as OpenCV does) -> rotation vector (3,)."""
    M = np.ascontiguousarray(np.asarray(R, np.float64).reshape(9))
    r = np.empty(3)
    _lib.check(_lib.lib().sfm_host_rodrigues(M.ctypes.data_as(_vp), 1, r.ctypes.data_as(_vp), None), "sfm_host_rodrigues")
    return r


def five_point(x1n, x2n):
    """Essential matrices consistent with five K-normalised correspondences (`sfm_host_five_point`): (m, 3, 3), m <= 10."""
    a = np.ascontiguousarray(np.asarray(x1n, np.float64).reshape(5, 2))
    b = np.ascontiguousarray(np.asarray(x2n, np.float64).reshape(5, 2))
    E, cnt = np.empty((10, 9)), np.zeros(1, np.int32)
    _lib.check(_lib.lib().sfm_host_five_point(a.ctypes.data_as(_vp), b.ctypes.data_as(_vp), E.ctypes.data_as(_vp), cnt.ctypes.data_as(_vp)),
               "sfm_host_five_point")
    return E[:cnt[0]].reshape(-1, 3, 3).copy()


def decompose_essential(E):
    """decomposeEssentialMat (`sfm_host_decompose_essential`): R1, R2, t."""
    Ec = np.ascontiguousarray(np.asarray(E, np.float64).reshape(9))
    R1, R2, t = np.empty((3, 3)), np.empty((3, 3)), np.empty(3)
    _lib.check(_lib.lib().sfm_host_decompose_essential(Ec.ctypes.data_as(_vp), R1.ctypes.data_as(_vp), R2.ctypes.data_as(_vp),
                                                       t.ctypes.data_as(_vp)), "sfm_host_decompose_essential")
    return R1, R2, t


def epnp(K, Xw, uv):
    """EPnP on 4..64 correspondences (`sfm_host_epnp`): R (3,3), t (3,)."""
    Kc = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
    X = np.ascontiguousarray(np.asarray(Xw, np.float64).reshape(-1, 3))
    u = np.ascontiguousarray(np.asarray(uv, np.float64).reshape(-1, 2))
    R, t = np.empty(9), np.empty(3)
    _lib.check(_lib.lib().sfm_host_epnp(Kc.ctypes.data_as(_vp), X.ctypes.data_as(_vp), u.ctypes.data_as(_vp), len(X),
                                        R.ctypes.data_as(_vp), t.ctypes.data_as(_vp)), "sfm_host_epnp")
    return R.reshape(3, 3), t


def p3p(K, Xw, uv):
    """solvePnP(P3P) on exactly four correspondences (`sfm_host_p3p`): (ok, R (3,3), t (3,))."""
    Kc = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
    X = np.ascontiguousarray(np.asarray(Xw, np.float64).reshape(4, 3))
    u = np.ascontiguousarray(np.asarray(uv, np.float64).reshape(4, 2))
    R, t, ok = np.zeros(9), np.zeros(3), np.zeros(1, np.int32)
    _lib.check(_lib.lib().sfm_host_p3p(Kc.ctypes.data_as(_vp), X.ctypes.data_as(_vp), u.ctypes.data_as(_vp),
                                       R.ctypes.data_as(_vp), t.ctypes.data_as(_vp), ok.ctypes.data_as(_vp)), "sfm_host_p3p")
    return bool(ok[0]), R.reshape(3, 3), t


def pnp_dlt_init(K, Xw, uv):
    """Non-planar initialisation of solvePnP(ITERATIVE) (`sfm_host_pnp_dlt_init`): (status, rvec, tvec);
    status 0 ok, 1 planar object, 2 fewer than 6 points."""
    Kc = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
    X = np.ascontiguousarray(np.asarray(Xw, np.float64).reshape(-1, 3))
    u = np.ascontiguousarray(np.asarray(uv, np.float64).reshape(-1, 2))
    r, t, st = np.zeros(3), np.zeros(3), np.zeros(1, np.int32)
    _lib.check(_lib.lib().sfm_host_pnp_dlt_init(Kc.ctypes.data_as(_vp), X.ctypes.data_as(_vp), u.ctypes.data_as(_vp), len(X),
                                                r.ctypes.data_as(_vp), t.ctypes.data_as(_vp), st.ctypes.data_as(_vp)), "sfm_host_pnp_dlt_init")
    return int(st[0]), r, t
