"""cv2-named facade over the HIP back-end: exactly the cv2 symbols sfm.py's hot path uses
(SURVEY §2.2 ★ rows), with cv2's shapes, dtypes and return conventions, so that

    from sfm_mvs_amd import cv2compat as cv2

makes sfm.py's helper functions (Triangulation sfm.py:45, PnP :60, ReprojectionError :79, the matcher
part of find_features :259-268, findEssentialMat/recoverPose :307-311) run on the MI355X unchanged.
NumPy in, NumPy out; every call uploads, launches the kernels of libsfmhip.so and downloads.
Also provided: `SIFT_create` / `xfeatures2d.SIFT_create` (`detectAndCompute`), `cvtColor(BGR2GRAY)`,
`pyrDown`, `KeyPoint`, and `imread` (host-side decode through PIL: file I/O, not a kernel).  Not provided (out of scope,
DESIGN.md §6): GUI.
"""
import numpy as np
import torch

from . import hostgeom as _hg
from . import ops as _ops
from . import ransac as _ransac
from . import sift as _sift

RANSAC = 8
NORM_L2 = 4
SOLVEPNP_ITERATIVE = 0
COLOR_BGR2GRAY = 6
TRIANGULATE_ROWS = 4        # 4: current OpenCV DLT system; 6: legacy cvTriangulatePoints (see docs/oracle.md)


def _dev():
    if not torch.cuda.is_available():
        raise _ops.SfmHipError("cv2compat needs a HIP device: there is no CPU fallback")
    return torch.device("cuda")


class DMatch:
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx=-1, trainIdx=-1, imgIdx=0, distance=float("inf")):
        self.queryIdx, self.trainIdx, self.imgIdx, self.distance = queryIdx, trainIdx, imgIdx, distance

    def __repr__(self):
        return f"DMatch(q={self.queryIdx}, t={self.trainIdx}, d={self.distance})"


class BFMatcher:
    """cv2.BFMatcher(normType=NORM_L2, crossCheck=False) — sfm.py:259, isfm.py:47."""

    def __init__(self, normType=NORM_L2, crossCheck=False):
        if normType != NORM_L2 or crossCheck:
            raise NotImplementedError("only BFMatcher(NORM_L2, crossCheck=False) is on the reference's path")

    def knnMatchArrays(self, queryDescriptors, trainDescriptors, k=2):
        """Array form: (trainIdx (nq,2) int32, distance (nq,2) float32)."""
        if k != 2:
            raise NotImplementedError("k=2 only (sfm.py:260)")
        d = _dev()
        q = torch.as_tensor(np.ascontiguousarray(queryDescriptors, np.float32)).to(d)
        t = torch.as_tensor(np.ascontiguousarray(trainDescriptors, np.float32)).to(d)
        idx, dist = _ops.knn2(q, t)
        return idx.cpu().numpy(), dist.cpu().numpy()

    def knnMatch(self, queryDescriptors, trainDescriptors, k=2):
        """list[nq] of [DMatch, DMatch] exactly as cv2 returns it (fewer entries when nt < k)."""
        idx, dist = self.knnMatchArrays(queryDescriptors, trainDescriptors, k)
        out = []
        for qi in range(idx.shape[0]):
            out.append([DMatch(qi, int(idx[qi, j]), 0, float(dist[qi, j])) for j in range(2) if idx[qi, j] >= 0])
        return out


def triangulatePoints(projMatr1, projMatr2, projPoints1, projPoints2):
    """cv2.triangulatePoints: (2,N) points (any strides, float32/float64) → (4,N) float32 (sfm.py:53)."""
    d = _dev()
    a = torch.as_tensor(np.asarray(projPoints1, np.float32)).to(d)
    b = torch.as_tensor(np.asarray(projPoints2, np.float32)).to(d)
    if a.dim() != 2 or a.shape[0] != 2:
        raise ValueError("triangulatePoints expects 2xN point arrays")
    return _ops.triangulate(np.asarray(projMatr1, np.float64), np.asarray(projMatr2, np.float64), a, b,
                            rows=TRIANGULATE_ROWS).cpu().numpy()


def Rodrigues(src):
    """cv2.Rodrigues: 3-vector → (3x3, jacobian=None) or 3x3 → ((3,1) vector, None)."""
    src = np.asarray(src, np.float64)
    if src.size == 9:
        return _hg.rodrigues_mat2vec(src.reshape(3, 3)).reshape(3, 1), None
    return _hg.rodrigues_vec2mat(src.reshape(3)), None


def convertPointsFromHomogeneous(src):
    """(N,4)/(N,3) → (N,1,3)/(N,1,2): divide by the last coordinate (w is already 1 at sfm.py:86,351)."""
    src = np.asarray(src)
    w = src[:, -1:]
    scale = np.where(w != 0, 1.0 / np.where(w != 0, w, 1), 1.0).astype(src.dtype)
    return (src[:, :-1] * scale).reshape(src.shape[0], 1, src.shape[1] - 1)


def projectPoints(objectPoints, rvec, tvec, cameraMatrix, distCoeffs=None):
    """cv2.projectPoints without distortion (sfm.py:88,121) → ((N,1,2) in the object points' dtype, None)."""
    if distCoeffs is not None and np.any(np.asarray(distCoeffs) != 0):
        raise NotImplementedError("the reference passes no distortion (sfm.py:88, :325)")
    d = _dev()
    X = np.asarray(objectPoints)
    out_dtype = np.float32 if X.dtype == np.float32 else np.float64
    Xf = torch.as_tensor(np.ascontiguousarray(X.reshape(-1, 3), np.float32)).to(d)
    cams = torch.as_tensor(np.hstack([np.asarray(rvec, np.float64).ravel(), np.asarray(tvec, np.float64).ravel()])[None]).to(d)
    obs = torch.zeros((Xf.shape[0], 2), dtype=torch.float32, device=d)
    out = _ops.project_residual(cams, np.asarray(cameraMatrix, np.float64), Xf, obs, want_proj=True)
    return out["proj"].cpu().numpy().astype(out_dtype).reshape(-1, 1, 2), None


def norm(src1, src2=None, normType=NORM_L2):
    """cv2.norm(a, b, NORM_L2): differences in the inputs' dtype, squares accumulated in double (sfm.py:93,95)."""
    if normType != NORM_L2:
        raise NotImplementedError("NORM_L2 only")
    a = np.asarray(src1)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    d = _dev()
    ta = torch.as_tensor(np.ascontiguousarray(a)).to(d)
    tb = None if src2 is None else torch.as_tensor(np.ascontiguousarray(np.asarray(src2), a.dtype)).to(d)
    return float(_ops.norm_l2(ta, tb).item())          # sfm_norm_l2: difference in the inputs' type, double accumulation


def findEssentialMat(points1, points2, cameraMatrix, method=RANSAC, prob=0.999, threshold=1.0, mask=None):
    if method != RANSAC:
        raise NotImplementedError("method=cv2.RANSAC only (sfm.py:307)")
    return _ransac.find_essential_mat(points1, points2, np.asarray(cameraMatrix, np.float64), prob, threshold)


def recoverPose(E, points1, points2, cameraMatrix):
    return _ransac.recover_pose(E, points1, points2, np.asarray(cameraMatrix, np.float64))


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs, rvec=None, tvec=None, useExtrinsicGuess=False,
                   iterationsCount=100, reprojectionError=8.0, confidence=0.99, inliers=None, flags=SOLVEPNP_ITERATIVE):
    """The reference calls this with cv2.SOLVEPNP_ITERATIVE as the 5th POSITIONAL argument (sfm.py:67), which
    is the `rvec` slot: it is ignored, exactly as OpenCV ignores it when useExtrinsicGuess is False."""
    if useExtrinsicGuess or flags != SOLVEPNP_ITERATIVE:
        raise NotImplementedError("defaults only (sfm.py:67)")
    if distCoeffs is not None and np.any(np.asarray(distCoeffs) != 0):
        raise NotImplementedError("zero distortion only (sfm.py:325,362)")
    return _ransac.solve_pnp_ransac(objectPoints, imagePoints, cameraMatrix, iterationsCount, reprojectionError, confidence)


class KeyPoint:
    """cv2.KeyPoint: .pt (x, y), .size, .angle, .response, .octave, .class_id — sfm.py:267-268 reads `.pt`."""
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x=0.0, y=0.0, size=0.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
        self.pt, self.size, self.angle, self.response, self.octave, self.class_id = (x, y), size, angle, response, octave, class_id

    def __repr__(self):
        return f"KeyPoint(pt={self.pt}, size={self.size}, angle={self.angle})"


def imread(filename, flags=1):
    """cv2.imread(path) (sfm.py:301-302,343): decoded on the HOST with PIL into cv2's layout — (H, W, 3) uint8 in B, G, R
    order (flags=1, the default) or (H, W) grey (flags=0) — and None when the file cannot be read, as cv2 does.  File
    decoding is I/O in front of the path, not a kernel; libjpeg builds may differ from OpenCV's in the last
    bit of a pixel."""
    try:
        from PIL import Image
        with Image.open(filename) as im:
            if flags == 0:
                return np.ascontiguousarray(np.asarray(im.convert("L"), np.uint8))
            return np.ascontiguousarray(np.asarray(im.convert("RGB"), np.uint8)[:, :, ::-1])
    except (OSError, ValueError):
        return None


def cvtColor(src, code):
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) on uint8 — sfm.py:243-244."""
    if code != COLOR_BGR2GRAY:
        raise NotImplementedError("only COLOR_BGR2GRAY is on the reference's path")
    return _sift.bgr2gray(torch.as_tensor(np.ascontiguousarray(src, np.uint8)).to(_dev())).cpu().numpy()


def pyrDown(src):
    """cv2.pyrDown(img) on uint8 — sfm.py:40 (`img_downscale`)."""
    return _sift.pyrdown(torch.as_tensor(np.ascontiguousarray(src, np.uint8)).to(_dev())).cpu().numpy()


class _Sift:
    """cv2.xfeatures2d.SIFT_create() / cv2.SIFT_create() — sfm.py:246."""

    def __init__(self, nfeatures=0, nOctaveLayers=3, contrastThreshold=0.04, edgeThreshold=10, sigma=1.6):
        if nfeatures != 0:
            raise NotImplementedError("nfeatures=0 only (sfm.py:246 uses the defaults)")
        self._params = (nOctaveLayers, contrastThreshold, edgeThreshold, sigma)
        self._engines = {}

    def detectAndComputeArrays(self, image):
        """Array form: (keypoints (n, 8) float32 {x, y, size, angle, response, octave bits, class_id bits, 0}, descriptors (n, 128) float32)."""
        img = np.ascontiguousarray(image)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise _ops.SfmHipError("detectAndCompute expects a single-channel uint8 image (sfm.py:243-244 converts first)")
        h, w = img.shape
        eng = self._engines.get((w, h))
        if eng is None:
            nl, ct, et, sg = self._params
            eng = self._engines[(w, h)] = _sift.Sift(w, h, _dev(), nl, ct, et, sg)
        kp, des = eng.run(torch.as_tensor(img).to(_dev()))
        return kp.cpu().numpy(), des.cpu().numpy()

    def detectAndCompute(self, image, mask=None):
        """-> (list of KeyPoint, descriptors (n, 128) float32) — sfm.py:247,252."""
        if mask is not None:
            raise NotImplementedError("mask=None only (sfm.py:247)")
        kp, des = self.detectAndComputeArrays(image)
        octv, cid = kp[:, 5].view(np.int32), kp[:, 6].view(np.int32)
        kps = [KeyPoint(float(k[0]), float(k[1]), float(k[2]), float(k[3]), float(k[4]), int(o), int(c)) for k, o, c in zip(kp, octv, cid)]
        return kps, des


def SIFT_create(*args, **kwargs):
    return _Sift(*args, **kwargs)


class xfeatures2d:      # namespace, as in opencv-contrib (sfm.py:246)
    SIFT_create = staticmethod(SIFT_create)
