"""Device-tensor operators of the hot path: thin, validating wrappers over libsfmhip.so.

Every function takes/returns torch tensors that already live in HBM and enqueues work on the
current HIP stream; nothing here computes on the host.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import SfmHipError, check, on_device, ptr, require_cuda, stream_ptr

_ws_cache = {}

# `filter` argument of the KNN entry points (include/sfm_hip.h SFM_KNN_FILTER_*): which candidate filter runs before the
# exact refine.  Results are bit-identical whichever runs.
KNN_FILTERS = {"auto": 0, "f32": 1, "split": 2, "lds": 3, "lds_split": 4, "half": 5, "noquant": 6}


def _filter_code(name):
    try:
        return KNN_FILTERS[name]
    except KeyError:
        raise SfmHipError(f"unknown KNN filter variant {name!r} (one of {sorted(KNN_FILTERS)})") from None


def _workspace(device, nbytes):
    """Grow-only scratch buffer per (device, current stream) — the C-ABI never allocates.  Work enqueued on different
    streams therefore never shares scratch; within a stream, calls are ordered.  (A buffer that has to grow is replaced:
    the caching allocator keeps the old block alive for the kernels already enqueued on its stream.)"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (device.type, idx, _lib.raw_stream(idx))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _f64_host(a, n, what):
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    if arr.size != n:
        raise SfmHipError(f"{what}: expected {n} values, got {arr.size}")
    return arr


def knn2(des0, des1, return_stats=False, filter="auto"):
    """cv2.BFMatcher().knnMatch(des0, des1, k=2) on device (sfm.py:259-260).

    des0 [nq,128] / des1 [nt,128] float32 CUDA tensors (rows may be strided).
    Returns idx [nq,2] int32 (trainIdx), dist [nq,2] float32 (DMatch.distance).
    filter: 'auto' (default) | 'f32' | 'split' | 'lds' | 'lds_split' | 'half' | 'noquant' — see KNN_FILTERS; identical results.
    """
    fcode = _filter_code(filter)
    require_cuda(des0, des1)
    if des0.dtype != torch.float32 or des1.dtype != torch.float32:
        raise SfmHipError("knn2: descriptors must be float32 (cv2 SIFT descriptors are CV_32F)")
    if des0.dim() != 2 or des1.dim() != 2 or des0.shape[1] != des1.shape[1]:
        raise SfmHipError("knn2: expected [nq,dim] and [nt,dim]")
    if des0.stride(1) != 1:
        des0 = des0.contiguous()
    if des1.stride(1) != 1:
        des1 = des1.contiguous()
    nq, dim = des0.shape
    nt = des1.shape[0]
    lib = _lib.lib()
    idx = torch.empty((nq, 2), dtype=torch.int32, device=des0.device)
    dist = torch.empty((nq, 2), dtype=torch.float32, device=des0.device)
    stats = torch.zeros(4, dtype=torch.int32, device=des0.device) if return_stats else None
    need = lib.sfm_knn2_l2_f32_ws_bytes(nq, nt, dim, fcode)
    if need == 0 and nq > 0:
        raise SfmHipError(f"knn2: unsupported shape nq={nq} nt={nt} dim={dim} (dim must be 128)")
    ws = _workspace(des0.device, need)
    ldq = des0.stride(0) if nq > 1 else dim
    ldt = des1.stride(0) if nt > 1 else dim
    with on_device(des0.device):
        check(lib.sfm_knn2_l2_f32(ptr(des0), nq, ldq, ptr(des1), nt, ldt, dim, fcode, ptr(idx), ptr(dist), ptr(stats),
                                  ptr(ws), ws.numel(), stream_ptr()), "sfm_knn2_l2_f32")
    return (idx, dist, stats) if return_stats else (idx, dist)


def ratio_compact(idx, dist, ratio=0.70, want_mask=False):
    """Lowe ratio test `m.distance < ratio * n.distance` + ordered compaction (sfm.py:262-265).

    Returns (query_idx [nq] int32, train_idx [nq] int32, count [1] int32 device scalar[, mask]).
    Only the first `count` entries are meaningful; order is ascending queryIdx.
    """
    require_cuda(idx, dist)
    nq = idx.shape[0]
    idx = idx.contiguous()
    dist = dist.contiguous()
    out_q = torch.empty(nq, dtype=torch.int32, device=idx.device)
    out_t = torch.empty(nq, dtype=torch.int32, device=idx.device)
    count = torch.zeros(1, dtype=torch.int32, device=idx.device)
    mask = torch.empty(nq, dtype=torch.uint8, device=idx.device) if want_mask else None
    lib = _lib.lib()
    rws = torch.empty(max(lib.sfm_ratio_compact_ws_bytes(nq), 256), dtype=torch.uint8, device=idx.device)
    with on_device(idx.device):
        check(lib.sfm_ratio_compact(ptr(idx), ptr(dist), nq, float(ratio), ptr(out_q), ptr(out_t), ptr(count), ptr(mask),
                                    ptr(rws), rws.numel(), stream_ptr()), "sfm_ratio_compact")
    return (out_q, out_t, count, mask) if want_mask else (out_q, out_t, count)


def gather_matches(kp0, kp1, out_q, out_t, count):
    """pts0 = kp0[queryIdx], pts1 = kp1[trainIdx] for the ratio survivors (sfm.py:267-268)."""
    require_cuda(kp0, kp1, out_q, out_t, count)
    kp0 = kp0.contiguous().float()
    kp1 = kp1.contiguous().float()
    cap = out_q.shape[0]
    pts0 = torch.empty((cap, 2), dtype=torch.float32, device=kp0.device)
    pts1 = torch.empty((cap, 2), dtype=torch.float32, device=kp0.device)
    with on_device(kp0.device):
        check(_lib.lib().sfm_gather_matches(ptr(kp0), ptr(kp1), ptr(out_q), ptr(out_t), ptr(count), cap, ptr(pts0),
                                            ptr(pts1), stream_ptr()), "sfm_gather_matches")
    return pts0, pts1


def common_points(pts1, pts2, raw=False):
    """sfm.py:215-239 association on device: (indx1 int32[m], indx2 int32[m], keep2 bool[n2]) — x OR y bit-equality,
    first hit, ascending order.  raw=True: no host read — (indx1 [n1] unsliced, indx2 [n1] unsliced, keep2 uint8 [n2], count int32 [1] on
    the device); the caller slices after ONE download of all the counts it is waiting for."""
    require_cuda(pts1, pts2)
    pts1 = pts1.contiguous().float().reshape(-1, 2)
    pts2 = pts2.contiguous().float().reshape(-1, 2)
    n1, n2 = pts1.shape[0], pts2.shape[0]
    dev = pts1.device
    first = torch.empty(max(n1, 1), dtype=torch.int32, device=dev)
    idx1 = torch.empty(max(n1, 1), dtype=torch.int32, device=dev)
    idx2 = torch.empty(max(n1, 1), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    keep2 = torch.empty(max(n2, 1), dtype=torch.uint8, device=dev)
    with on_device(dev):
        check(_lib.lib().sfm_common_points(ptr(pts1), n1, ptr(pts2), n2, ptr(first), ptr(idx1), ptr(idx2), ptr(count),
                                           ptr(keep2), stream_ptr()), "sfm_common_points")
    if raw:
        return idx1, idx2, keep2[:n2], count
    m = int(count.item())
    return idx1[:m], idx2[:m], keep2[:n2].bool()


def mask_indices(mask, nonzero=False, raw=False):
    """Rows of a uint8 mask that pass, ascending, as an int32 CUDA tensor (sfm_mask_indices): `mask.ravel() == 1` (OpenCV's
    {0,1} essential-matrix mask, sfm.py:309) or, nonzero=True, `mask.ravel() > 0` (the {0,255} cheirality mask, sfm.py:313; the
    complement of common_points).  One host read (the count sizes the result)."""
    require_cuda(mask)
    m = mask.reshape(-1)
    if m.dtype == torch.bool:
        m = m.view(torch.uint8)
    if m.dtype != torch.uint8:
        raise SfmHipError("mask_indices: uint8 / bool mask")
    m = m.contiguous()
    n = m.numel()
    out = torch.empty(max(n, 1), dtype=torch.int32, device=m.device)
    count = torch.zeros(1, dtype=torch.int32, device=m.device)
    lib = _lib.lib()
    ws = _workspace(m.device, lib.sfm_mask_indices_ws_bytes(n))
    with on_device(m.device):
        check(lib.sfm_mask_indices(ptr(m), n, 1 if nonzero else 0, ptr(out), ptr(count), ptr(ws), ws.numel(), stream_ptr()), "sfm_mask_indices")
    if raw:
        return out, count
    return out[:int(count.item())]


def knn_merge_top2(cand):
    """Merge the shards' partial 2-NN results (train set split over devices): cand [S][2][nq][2] int32 CUDA tensor — per
    shard the global trainIdx plane and the distance-bits plane — -> (idx [nq,2] int32, dist [nq,2] float32), ordered by
    (distance, trainIdx) as a single scan would (sfm_knn_merge_top2)."""
    require_cuda(cand)
    if cand.dtype != torch.int32 or cand.dim() != 4 or cand.shape[1] != 2 or cand.shape[3] != 2:
        raise SfmHipError("knn_merge_top2: cand must be int32 [S][2][nq][2]")
    cand = cand.contiguous()
    S, nq = int(cand.shape[0]), int(cand.shape[2])
    idx = torch.empty((nq, 2), dtype=torch.int32, device=cand.device)
    dist = torch.empty((nq, 2), dtype=torch.float32, device=cand.device)
    with on_device(cand.device):
        check(_lib.lib().sfm_knn_merge_top2(ptr(cand), S, nq, ptr(idx), ptr(dist), stream_ptr()), "sfm_knn_merge_top2")
    return idx, dist


def match_pair(des0, des1, ratio=0.70, filter="auto"):
    """KNN + ratio for one image pair; returns (query_idx, train_idx, dist1) trimmed to the survivors.

    One host sync (reading the survivor count) — this is the find_features() boundary.
    """
    require_cuda(des0, des1)
    if des0.dim() != 2 or des1.dim() != 2 or des0.shape[1] != 128 or des1.shape[1] != 128:
        raise SfmHipError("match_pair: descriptors must be [n,128]")
    if des0.dtype != torch.float32 or des1.dtype != torch.float32:
        raise SfmHipError("match_pair: descriptors must be float32 (cv2 SIFT output)")
    pm = PairMatcher(des0.shape[0], des1.shape[0], des0.device, ratio, filter=filter)
    idx, dist, out_q, out_t, count = pm.run(des0 if des0.stride(1) == 1 else des0.contiguous(),
                                            des1 if des1.stride(1) == 1 else des1.contiguous())
    m = int(count.item())
    return out_q[:m], out_t[:m], idx, dist


def triangulate(P1, P2, pts1, pts2, rows=4, normalise_w=False):
    """cv2.triangulatePoints on device (sfm.py:53) [+ `cloud / cloud[3]` (sfm.py:54)].

    P1, P2: 3x4 host matrices (float64).  pts1/pts2: float32 CUDA tensors shaped (2,N) like cv2's
    argument — any strides, so the reference's transposed views of (N,2) arrays work unchanged.
    Returns X4 (4,N) float32 CUDA tensor.  normalise_w: False (raw singular vector, OpenCV's sign), True (divided by
    w in float32), "fast" (the same normalised result via inverse iteration on A^T A instead of Jacobi sweeps:
    bit-identical on > 99.9 % of points, 1 ulp otherwise; rows = 4 only) or "guarded" (the fast path where its float32
    casts equal the faithful path's under an EMPIRICAL bound (|fast - Jacobi| <= 2.5 sens measured on 4e6 points, guard at 16 sens; bench.py and tests/test_gpu_geometry.py bit-compare against normalise_w = 1), the Jacobi sweeps — compacted — for the ~2 % of points near a rounding
    boundary: bit-identical to True, ~8x faster; what the driver uses).
    """
    require_cuda(pts1, pts2)
    if normalise_w == "fast":
        normalise_w = 2
    elif normalise_w == "guarded":
        normalise_w = 3
    if pts1.dtype != torch.float32 or pts2.dtype != torch.float32:
        raise SfmHipError("triangulate: points must be float32")
    if pts1.dim() != 2 or pts1.shape[0] != 2 or pts2.shape != pts1.shape:
        raise SfmHipError("triangulate: expected two (2,N) arrays")
    n = pts1.shape[1]
    if pts2.stride() != pts1.stride():
        pts2 = pts2.contiguous()
        pts1 = pts1.contiguous()
    p1 = _f64_host(P1, 12, "P1")
    p2 = _f64_host(P2, 12, "P2")
    X4 = torch.empty((4, n), dtype=torch.float32, device=pts1.device)
    spt, sxy = (pts1.stride(1), pts1.stride(0)) if n > 0 else (1, 1)
    with on_device(pts1.device):
        check(_lib.lib().sfm_triangulate_dlt(p1.ctypes.data_as(ctypes.c_void_p), p2.ctypes.data_as(ctypes.c_void_p),
                                             ptr(pts1), ptr(pts2), n, spt, sxy, int(rows), int(normalise_w),
                                             ptr(X4), stream_ptr()), "sfm_triangulate_dlt")
    return X4


def triangulate_matches_batch(knn_blocks, n_query, keypoints0, keypoints1, proj0, proj1, out, counts=None, ratio=0.70):
    """Lowe loop + keypoint gather + Triangulation (sfm.py:262-268, :53-54) of 1..8 pairs in one set of launches, straight
    from the KNN blocks to point blocks (sfm_triangulate_matches_batch) — no host round trip, no survivor list.

    knn_blocks: per pair an int32 [2][cap_b][2] CUDA tensor ([0] trainIdx x2, [1] float32 distance bits x2: what
    PairMatcher / BatchMatcher write and the pair-sharded exchange gathers); n_query: queries per pair;
    keypoints0/1: per pair [n,2] float32 CUDA (KeyPoint.pt of the query / train image); proj0/1: per pair 3x4 float64;
    out: per pair a float32 [4][cap] CUDA view (contiguous rows, one cap for all) receiving the points, columns >= the
    survivor count zeroed; counts: optional per pair int32 [1] CUDA views receiving the survivor count.
    Values are those of ops.triangulate(..., normalise_w="guarded") on the gathered survivors, bit for bit."""
    B = len(knn_blocks)
    if not (1 <= B <= 8) or not (len(n_query) == len(keypoints0) == len(keypoints1) == len(proj0) == len(proj1) == len(out) == B):
        raise SfmHipError("triangulate_matches_batch: 1..8 pairs, one entry per pair in every argument")
    cap = int(out[0].shape[1])
    vp = ctypes.c_void_p
    for b in range(B):
        kb, o = knn_blocks[b], out[b]
        require_cuda(kb, keypoints0[b], keypoints1[b], o)
        if kb.dtype != torch.int32 or kb.dim() != 3 or kb.shape[0] != 2 or kb.shape[2] != 2 or kb.stride(2) != 1 or kb.stride(1) != 2:
            raise SfmHipError("triangulate_matches_batch: a KNN block is an int32 [2][cap][2] tensor with contiguous planes")
        if int(n_query[b]) > min(cap, kb.shape[1]):
            raise SfmHipError("triangulate_matches_batch: more queries than block rows / output columns")
        if o.dtype != torch.float32 or tuple(o.shape) != (4, cap) or o.stride(1) != 1 or o.stride(0) != cap:
            raise SfmHipError("triangulate_matches_batch: out must be float32 [4][cap] views with one cap and contiguous rows")
        for kp in (keypoints0[b], keypoints1[b]):
            if kp.dtype != torch.float32 or kp.dim() != 2 or kp.shape[1] != 2 or not kp.is_contiguous():
                raise SfmHipError("triangulate_matches_batch: keypoints must be contiguous float32 [n,2]")
        if counts is not None:
            require_cuda(counts[b])
    arr = lambda ptrs: (vp * B)(*ptrs)
    P = np.empty((B, 2, 12), dtype=np.float64)
    for b in range(B):
        P[b, 0] = _f64_host(proj0[b], 12, "P1")
        P[b, 1] = _f64_host(proj1[b], 12, "P2")
    nq = np.asarray([int(n) for n in n_query], dtype=np.int64)
    dev = out[0].device
    lib = _lib.lib()
    ws = _workspace(dev, lib.sfm_triangulate_matches_batch_ws_bytes(B, cap))
    with on_device(dev):
        check(lib.sfm_triangulate_matches_batch(B, arr([k[0].data_ptr() for k in knn_blocks]), arr([k[1].data_ptr() for k in knn_blocks]),
                                                nq.ctypes.data_as(vp), float(ratio), arr([k.data_ptr() for k in keypoints0]),
                                                arr([k.data_ptr() for k in keypoints1]), P.ctypes.data_as(vp), cap,
                                                arr([o.data_ptr() for o in out]),
                                                None if counts is None else arr([c.data_ptr() for c in counts]),
                                                ptr(ws), ws.numel(), stream_ptr()), "sfm_triangulate_matches_batch")


def project_residual(cams, K, X, obs, cam_idx=None, pt_idx=None, thr2=64.0, want_proj=True, want_inlier=False,
                     want_jac=False, want_pt_jac=False, want_res2=False):
    """Reprojection sweep (sfm.py:79-100 / :67 scoring / :104-136 residual).

    cams [ncam,6] float64 CUDA (rvec, tvec); K 3x3 host; X [npt,3] float32 CUDA; obs [nobs,2] float32 CUDA.
    Returns a dict of CUDA tensors: proj, sumsq (1,), inlier, JtJ_cam, Jtr_cam, JtJ_pt, Jtr_pt (as requested).
    """
    require_cuda(cams, X, obs, cam_idx, pt_idx)
    cams = cams.contiguous().to(torch.float64).reshape(-1, 6)
    if X.dtype != torch.float32 or obs.dtype != torch.float32:
        raise SfmHipError("project_residual: X and obs must be float32")
    if X.stride(-1) != 1:
        X = X.contiguous()
    obs = obs.contiguous().reshape(-1, 2)
    ncam, npt, nobs = cams.shape[0], X.shape[0], obs.shape[0]
    dev = X.device
    k = _f64_host(K, 9, "K")
    out = {"sumsq": torch.zeros(1, dtype=torch.float64, device=dev)}
    if want_proj:
        out["proj"] = torch.empty((nobs, 2), dtype=torch.float32, device=dev)
    if want_inlier:
        out["inlier"] = torch.empty(nobs, dtype=torch.uint8, device=dev)
    if want_jac:
        out["JtJ_cam"] = torch.zeros((ncam, 36), dtype=torch.float64, device=dev)
        out["Jtr_cam"] = torch.zeros((ncam, 6), dtype=torch.float64, device=dev)
    if want_pt_jac:
        out["JtJ_pt"] = torch.zeros((npt, 9), dtype=torch.float64, device=dev)
        out["Jtr_pt"] = torch.zeros((npt, 3), dtype=torch.float64, device=dev)
    if want_res2:
        out["res2"] = torch.zeros(1, dtype=torch.float64, device=dev)
    lib = _lib.lib()
    ws = _workspace(dev, lib.sfm_project_residual_ws_bytes(nobs, ncam, npt))
    ci = None if cam_idx is None else cam_idx.contiguous().to(torch.int32)
    pi = None if pt_idx is None else pt_idx.contiguous().to(torch.int32)
    with on_device(dev):
        check(lib.sfm_project_residual(ptr(cams), ncam, k.ctypes.data_as(ctypes.c_void_p), ptr(X), npt,
                                       X.stride(0) if npt > 1 else 3, ptr(obs), ptr(ci), ptr(pi), nobs,
                                       ptr(out.get("proj")), ptr(out["sumsq"]), ptr(out.get("inlier")), float(thr2),
                                       ptr(out.get("JtJ_cam")), ptr(out.get("Jtr_cam")), ptr(out.get("JtJ_pt")),
                                       ptr(out.get("Jtr_pt")), ptr(out.get("res2")), ptr(ws), ws.numel(), stream_ptr()),
              "sfm_project_residual")
    return out


def ba_dense_sweep(cams, K, X, obs, want_cam=True, want_pt=True):
    """Dense-visibility residual / J^T J sweep (config 4).  obs [ncam,npt,2] float32 CUDA."""
    require_cuda(cams, X, obs)
    cams = cams.contiguous().to(torch.float64).reshape(-1, 6)
    ncam, npt = cams.shape[0], X.shape[0]
    if obs.dtype != torch.float32 or X.dtype != torch.float32 or tuple(obs.shape) != (ncam, npt, 2):
        raise SfmHipError("ba_dense_sweep: obs must be float32 [ncam,npt,2], X float32 [npt,3]")
    obs = obs.contiguous()
    if X.stride(-1) != 1:
        X = X.contiguous()
    dev = X.device
    k = _f64_host(K, 9, "K")
    out = {"sumsq": torch.zeros(1, dtype=torch.float64, device=dev)}
    if want_cam:
        out["JtJ_cam"] = torch.empty((ncam, 36), dtype=torch.float64, device=dev)
        out["Jtr_cam"] = torch.empty((ncam, 6), dtype=torch.float64, device=dev)
    if want_pt:
        out["JtJ_pt"] = torch.empty((npt, 9), dtype=torch.float64, device=dev)
        out["Jtr_pt"] = torch.empty((npt, 3), dtype=torch.float64, device=dev)
    lib = _lib.lib()
    ws = _workspace(dev, lib.sfm_ba_dense_sweep_ws_bytes(ncam, npt))
    with on_device(dev):
        check(lib.sfm_ba_dense_sweep(ptr(cams), ncam, k.ctypes.data_as(ctypes.c_void_p), ptr(X), npt, X.stride(0),
                                     ptr(obs), ptr(out["sumsq"]), ptr(out.get("JtJ_cam")), ptr(out.get("Jtr_cam")),
                                     ptr(out.get("JtJ_pt")), ptr(out.get("Jtr_pt")), ptr(ws), ws.numel(), stream_ptr()),
              "sfm_ba_dense_sweep")
    return out


def block_inverse(A, k, check_singular=True):
    """Inverse of n independent k x k float64 blocks ([n, k*k] or [n, k, k]; k = 3 or 6) on the device
    (sfm_block_inverse_checked).  check_singular=True (default) reads the device status word — one synchronisation — and
    raises if any block met a zero / non-finite pivot; False returns (inverse, bad-count tensor) without synchronising."""
    require_cuda(A)
    if A.dtype != torch.float64:
        raise SfmHipError("block_inverse: float64 blocks")
    a = A.contiguous()
    out = torch.empty_like(a)
    n = a.numel() // (k * k)
    bad = torch.zeros(1, dtype=torch.int32, device=a.device)
    with on_device(a.device):
        check(_lib.lib().sfm_block_inverse_checked(ptr(a), n, int(k), ptr(out), ptr(bad), stream_ptr()), "sfm_block_inverse_checked")
    if not check_singular:
        return out, bad
    nb = int(bad.item())
    if nb:
        raise SfmHipError(f"block_inverse: {nb} of {n} {k}x{k} blocks are singular or non-finite")
    return out


def block_matvec(A, x, k):
    """y_i = A_i x_i for n k x k float64 blocks and x [n, k] (sfm_block_matvec)."""
    require_cuda(A, x)
    if A.dtype != torch.float64 or x.dtype != torch.float64:
        raise SfmHipError("block_matvec: float64 operands")
    a, xv = A.contiguous(), x.contiguous()
    n = xv.numel() // k
    if a.numel() != n * k * k:
        raise SfmHipError("block_matvec: shape mismatch")
    y = torch.empty_like(xv)
    with on_device(a.device):
        check(_lib.lib().sfm_block_matvec(ptr(a), ptr(xv), n, int(k), ptr(y), stream_ptr()), "sfm_block_matvec")
    return y


def project_points_f64(rvec, tvec, K, X):
    """cv2.projectPoints on float64 object points [n,3] (CUDA) -> [n,2] float64 (sfm_project_points_f64)."""
    require_cuda(X)
    if X.dtype != torch.float64:
        raise SfmHipError("project_points_f64: float64 object points")
    Xc = X.contiguous().reshape(-1, 3)
    out = torch.empty((Xc.shape[0], 2), dtype=torch.float64, device=X.device)
    r, t, k = _f64_host(rvec, 3, "rvec"), _f64_host(tvec, 3, "tvec"), _f64_host(K, 9, "K")
    vp = ctypes.c_void_p
    with on_device(X.device):
        check(_lib.lib().sfm_project_points_f64(r.ctypes.data_as(vp), t.ctypes.data_as(vp), k.ctypes.data_as(vp), ptr(Xc), Xc.shape[0],
                                                ptr(out), stream_ptr()), "sfm_project_points_f64")
    return out


def norm_l2(a, b=None):
    """cv2.norm(a, b, NORM_L2) on device tensors of one dtype (float32 / float64): a [1] float64 device scalar."""
    require_cuda(a, b)
    if a.dtype not in (torch.float32, torch.float64) or (b is not None and (b.dtype != a.dtype or b.numel() != a.numel())):
        raise SfmHipError("norm_l2: float32 or float64 operands of equal size")
    av = a.contiguous()
    bv = b.contiguous() if b is not None else None
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    lib = _lib.lib()
    ws = _workspace(a.device, lib.sfm_norm_l2_ws_bytes())
    with on_device(a.device):
        check(lib.sfm_norm_l2(ptr(av), ptr(bv), av.numel(), 1 if a.dtype == torch.float64 else 0, ptr(out), ptr(ws), ws.numel(),
                              stream_ptr()), "sfm_norm_l2")
    return out


def _schur_args(cams, K, X):
    require_cuda(cams, X)
    cams = cams.contiguous().to(torch.float64).reshape(-1, 6)
    if X.dtype != torch.float32 or X.dim() != 2 or X.shape[1] < 3:
        raise SfmHipError("schur product: X must be float32 [npt,>=3]")
    if X.stride(-1) != 1:
        X = X.contiguous()
    return cams, _f64_host(K, 9, "K"), X


def _schur_indexed(cams, k, X, cam_idx, pt_idx, mode, vec):
    require_cuda(cam_idx, pt_idx)
    ncam, npt, dev = cams.shape[0], X.shape[0], X.device
    cam_idx = cam_idx.contiguous().to(torch.int32)
    pt_idx = pt_idx.contiguous().to(torch.int32)
    out = torch.empty((npt, 3) if mode == 0 else (ncam, 6), dtype=torch.float64, device=dev)
    lib = _lib.lib()
    ws = _workspace(dev, lib.sfm_ba_schur_indexed_ws_bytes(ncam))
    with on_device(dev):
        check(lib.sfm_ba_schur_indexed(ptr(cams), ncam, k.ctypes.data_as(ctypes.c_void_p), ptr(X), npt, X.stride(0), ptr(cam_idx),
                                       ptr(pt_idx), cam_idx.numel(), mode, ptr(vec), ptr(out), ptr(ws), ws.numel(), stream_ptr()),
              "sfm_ba_schur_indexed")
    return out


def ba_schur_wt(cams, K, X, x_cam, cam_idx=None, pt_idx=None):
    """u = W^T x: x_cam [ncam,6] float64 → u [npt,3] float64.  Dense visibility unless cam_idx / pt_idx (one entry per
    observation) are given.  No observation data is read."""
    cams, k, X = _schur_args(cams, K, X)
    ncam, npt, dev = cams.shape[0], X.shape[0], X.device
    x_cam = x_cam.contiguous().to(torch.float64).reshape(ncam, 6)
    if cam_idx is not None:
        return _schur_indexed(cams, k, X, cam_idx, pt_idx, 0, x_cam)
    u = torch.empty((npt, 3), dtype=torch.float64, device=dev)
    lib = _lib.lib()
    ws = _workspace(dev, lib.sfm_ba_schur_ws_bytes(ncam, npt))
    with on_device(dev):
        check(lib.sfm_ba_schur_wt(ptr(cams), ncam, k.ctypes.data_as(ctypes.c_void_p), ptr(X), npt, X.stride(0), ptr(x_cam), ptr(u),
                                  ptr(ws), ws.numel(), stream_ptr()), "sfm_ba_schur_wt")
    return u


def ba_schur_w(cams, K, X, v_pt, cam_idx=None, pt_idx=None):
    """w = W v: v_pt [npt,3] float64 → w [ncam,6] float64 (dense visibility unless cam_idx / pt_idx are given)."""
    cams, k, X = _schur_args(cams, K, X)
    ncam, npt, dev = cams.shape[0], X.shape[0], X.device
    v_pt = v_pt.contiguous().to(torch.float64).reshape(npt, 3)
    if cam_idx is not None:
        return _schur_indexed(cams, k, X, cam_idx, pt_idx, 1, v_pt)
    w = torch.empty((ncam, 6), dtype=torch.float64, device=dev)
    lib = _lib.lib()
    ws = _workspace(dev, lib.sfm_ba_schur_ws_bytes(ncam, npt))
    with on_device(dev):
        check(lib.sfm_ba_schur_w(ptr(cams), ncam, k.ctypes.data_as(ctypes.c_void_p), ptr(X), npt, X.stride(0), ptr(v_pt), ptr(w),
                                 ptr(ws), ws.numel(), stream_ptr()), "sfm_ba_schur_w")
    return w


def ba_schur_solve(cams, K, X, blocks, lam, fix_first_camera=True, cg_tol=1e-10, cg_iters=200):
    """One damped Schur-complement step of the dense problem, PCG on the device (sfm_ba_schur_solve).  `blocks`: the dict
    ops.ba_dense_sweep returned at (cams, X).  Returns (dc [ncam,6], dp [npt,3] float64 CUDA, CG iterations, status bits)."""
    cams, k, X = _schur_args(cams, K, X)
    ncam, npt, dev = cams.shape[0], X.shape[0], X.device
    B, gc, C, gp = (blocks[n].contiguous() for n in ("JtJ_cam", "Jtr_cam", "JtJ_pt", "Jtr_pt"))
    dc = torch.empty((ncam, 6), dtype=torch.float64, device=dev)
    dp = torch.empty((npt, 3), dtype=torch.float64, device=dev)
    lib = _lib.lib()
    ws = _workspace(dev, lib.sfm_ba_schur_solve_ws_bytes(ncam, npt))
    it, st = ctypes.c_int32(0), ctypes.c_int32(0)
    with on_device(dev):
        check(lib.sfm_ba_schur_solve(ptr(cams), ncam, k.ctypes.data_as(ctypes.c_void_p), ptr(X), npt, X.stride(0), ptr(B), ptr(gc), ptr(C), ptr(gp),
                                     float(lam), int(bool(fix_first_camera)), float(cg_tol), int(cg_iters), ptr(dc), ptr(dp),
                                     ctypes.byref(it), ctypes.byref(st), ptr(ws), ws.numel(), stream_ptr()), "sfm_ba_schur_solve")
    return dc, dp, it.value, st.value


def score_essential(E, x1n, x2n, thr2, want_mask=False):
    """Sampson-distance inlier counts for h candidate essential matrices (sfm.py:307 RANSAC scoring)."""
    require_cuda(E, x1n, x2n)
    E = E.contiguous().to(torch.float64).reshape(-1, 9)
    x1n = x1n.contiguous().to(torch.float64).reshape(-1, 2)
    x2n = x2n.contiguous().to(torch.float64).reshape(-1, 2)
    h, n = E.shape[0], x1n.shape[0]
    counts = torch.empty(h, dtype=torch.int32, device=E.device)
    mask = torch.empty((h, n), dtype=torch.uint8, device=E.device) if want_mask else None
    with on_device(E.device):
        check(_lib.lib().sfm_score_essential(ptr(E), h, ptr(x1n), ptr(x2n), n, float(thr2), ptr(counts), ptr(mask),
                                             stream_ptr()), "sfm_score_essential")
    return (counts, mask) if want_mask else counts


def score_pnp(poses, K, X, obs, thr2=64.0, want_mask=False):
    """Reprojection inlier counts for h PnP hypotheses (rvec,tvec) (sfm.py:67 RANSAC scoring)."""
    require_cuda(poses, X, obs)
    poses = poses.contiguous().to(torch.float64).reshape(-1, 6)
    X = X.contiguous().reshape(-1, 3)
    obs = obs.contiguous().reshape(-1, 2)
    if X.dtype != torch.float32 or obs.dtype != torch.float32:
        raise SfmHipError("score_pnp: X and obs must be float32")
    h, n = poses.shape[0], X.shape[0]
    k = _f64_host(K, 9, "K")
    counts = torch.empty(h, dtype=torch.int32, device=X.device)
    mask = torch.empty((h, n), dtype=torch.uint8, device=X.device) if want_mask else None
    with on_device(X.device):
        check(_lib.lib().sfm_score_pnp(ptr(poses), h, k.ctypes.data_as(ctypes.c_void_p), ptr(X), ptr(obs), n,
                                       float(thr2), ptr(counts), ptr(mask), stream_ptr()), "sfm_score_pnp")
    return (counts, mask) if want_mask else counts


class PairMatcher:
    """Pre-planned KNN + Lowe-ratio for a fixed (nq, nt): all outputs and the workspace are allocated
    once, so a call enqueues kernels only (no allocator traffic, no host sync).  This is the object the
    pair-sharded matcher and bench.py drive."""

    def __init__(self, nq, nt, device, ratio=0.70, dim=128, filter="auto"):
        self.nq, self.nt, self.dim, self.ratio = int(nq), int(nt), int(dim), float(ratio)
        self.device = torch.device(device)
        self.filter = _filter_code(filter)
        lib = _lib.lib()
        need = lib.sfm_match_l2_f32_ws_bytes(self.nq, self.nt, self.dim, self.filter)
        if need == 0 and self.nq > 0:
            raise SfmHipError(f"PairMatcher: unsupported shape nq={nq} nt={nt} dim={dim}")
        self.ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
        # idx and dist share one allocation ([2][nq][2] x 4 bytes): the pair-sharded exchange all-gathers `result` as is
        self.result = torch.empty((2, self.nq, 2), dtype=torch.int32, device=self.device)
        self.idx = self.result[0]
        self.dist = self.result[1].view(torch.float32)
        self.stats = torch.zeros(4, dtype=torch.int32, device=self.device)
        self.out_q = torch.empty(self.nq, dtype=torch.int32, device=self.device)
        self.out_t = torch.empty(self.nq, dtype=torch.int32, device=self.device)
        self.count = torch.zeros(1, dtype=torch.int32, device=self.device)

    def run(self, des0, des1, result=None):
        """result: optional int32 [2][nq][2] CUDA tensor to receive (trainIdx, distance bits) instead of self.result —
        e.g. a slot of the batch buffer the pair-sharded exchange all-gathers."""
        require_cuda(des0, des1)
        if tuple(des0.shape) != (self.nq, self.dim) or tuple(des1.shape) != (self.nt, self.dim):
            raise SfmHipError("PairMatcher.run: shape differs from the plan")
        idx, dist = self.idx, self.dist
        if result is not None:
            require_cuda(result)
            if tuple(result.shape) != (2, self.nq, 2) or result.dtype != torch.int32 or not result.is_contiguous():
                raise SfmHipError("PairMatcher.run: result must be a contiguous int32 [2][nq][2] tensor")
            idx, dist = result[0], result[1].view(torch.float32)
        if des0.dtype != torch.float32 or des1.dtype != torch.float32 or des0.stride(1) != 1 or des1.stride(1) != 1:
            raise SfmHipError("PairMatcher.run: float32 row-major descriptors required")
        with on_device(self.device):
            check(_lib.lib().sfm_match_l2_f32(ptr(des0), self.nq, des0.stride(0), ptr(des1), self.nt, des1.stride(0), self.dim, self.filter,
                                              self.ratio, ptr(idx), ptr(dist), ptr(self.out_q), ptr(self.out_t),
                                              ptr(self.count), None, ptr(self.stats), ptr(self.ws), self.ws.numel(), stream_ptr()),
                  "sfm_match_l2_f32")
        return idx, dist, self.out_q, self.out_t, self.count


class BatchMatcher:
    """KNN + Lowe ratio for up to `batch` pairs of one shape per call (sfm_match_batch_l2_f32): one prep, one filter, one
    refine and one scatter launch for the whole batch — the filter's workgroups amortise their prologue over `batch`
    times the work, and there are `batch` times fewer kernel boundaries.  Outputs and workspace are allocated once.
    Results of pair b: idx[b] [nq,2] int32, dist[b] [nq,2] float32, out_q[b], out_t[b] [nq] int32, count[b] [1] int32."""

    def __init__(self, nq, nt, device, ratio=0.70, batch=4, dim=128, filter="auto"):
        self.nq, self.nt, self.dim, self.ratio, self.batch = int(nq), int(nt), int(dim), float(ratio), int(batch)
        self.device = torch.device(device)
        self.filter = _filter_code(filter)
        lib = _lib.lib()
        need = lib.sfm_match_batch_l2_f32_ws_bytes(self.nq, self.nt, self.dim, self.batch, self.filter)
        if need == 0 and self.nq > 0:
            raise SfmHipError(f"BatchMatcher: unsupported configuration nq={nq} nt={nt} dim={dim} batch={batch}")
        self.ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
        B = self.batch
        self.result = torch.empty((B, 2, self.nq, 2), dtype=torch.int32, device=self.device)     # [b][0] trainIdx, [b][1] distance bits
        self.idx = self.result[:, 0]
        self.dist = self.result[:, 1].view(torch.float32)
        self.stats = torch.zeros((B, 4), dtype=torch.int32, device=self.device)
        self.out_q = torch.empty((B, self.nq), dtype=torch.int32, device=self.device)
        self.out_t = torch.empty((B, self.nq), dtype=torch.int32, device=self.device)
        self.count = torch.zeros((B, 1), dtype=torch.int32, device=self.device)
        vp = ctypes.c_void_p
        self._arr = lambda ptrs: (vp * len(ptrs))(*ptrs)
        self._fixed = {k: self._arr([t[b].data_ptr() for b in range(B)]) for k, t in
                       (("out_q", self.out_q), ("out_t", self.out_t), ("count", self.count), ("stats", self.stats))}
        self._own = (self._arr([self.result[b, 0].data_ptr() for b in range(B)]), self._arr([self.result[b, 1].data_ptr() for b in range(B)]))

    def run(self, pairs, results=None):
        """pairs: list of (des0, des1) CUDA tensors ([nq,128], [nt,128] float32 row-major), 1 <= len <= batch.
        results: optional list of int32 [2][nq][2] CUDA tensors receiving (trainIdx, distance bits) of each pair instead of
        self.result[b] — e.g. the slots of an exchange buffer.  Returns len(pairs)."""
        n = len(pairs)
        if not 1 <= n <= self.batch:
            raise SfmHipError(f"BatchMatcher.run: 1..{self.batch} pairs per call (got {n})")
        for d0, d1 in pairs:
            require_cuda(d0, d1)
            if tuple(d0.shape) != (self.nq, self.dim) or tuple(d1.shape) != (self.nt, self.dim):
                raise SfmHipError("BatchMatcher.run: shape differs from the plan")
            if d0.dtype != torch.float32 or d1.dtype != torch.float32 or d0.stride(1) != 1 or d1.stride(1) != 1:
                raise SfmHipError("BatchMatcher.run: float32 row-major descriptors required")
            if d0.stride(0) != pairs[0][0].stride(0) or d1.stride(0) != pairs[0][1].stride(0):
                raise SfmHipError("BatchMatcher.run: the pairs of a batch share their row strides")
        q = self._arr([p[0].data_ptr() for p in pairs])
        t = self._arr([p[1].data_ptr() for p in pairs])
        idx, dist = self._own
        if results is not None:
            if len(results) != n:                          # (a short pointer array would be read past its end by the library)
                raise SfmHipError(f"BatchMatcher.run: {n} pairs but {len(results)} result blocks")
            for r in results:
                require_cuda(r)
                if tuple(r.shape) != (2, self.nq, 2) or r.dtype != torch.int32 or not r.is_contiguous():
                    raise SfmHipError("BatchMatcher.run: result must be a contiguous int32 [2][nq][2] tensor")
            idx = self._arr([r[0].data_ptr() for r in results])
            dist = self._arr([r[1].data_ptr() for r in results])
        f = self._fixed
        with on_device(self.device):
            check(_lib.lib().sfm_match_batch_l2_f32(n, q, self.nq, pairs[0][0].stride(0), t, self.nt, pairs[0][1].stride(0), self.dim, self.filter,
                                                    self.ratio, idx, dist, f["out_q"], f["out_t"], f["count"], None, f["stats"],
                                                    ptr(self.ws), self.ws.numel(), stream_ptr()), "sfm_match_batch_l2_f32")
        return n


class PairPipeline:
    """Independent image pairs pipelined over `depth` HIP streams (one PairMatcher = one workspace + output set per
    stream).  A pair's step ends with low-occupancy phases — the refine kernel's few rescanning workgroups, the
    10-workgroup ordered scatter, the next pair's prep pass — that a single stream serialises; on separate streams
    they overlap the neighbouring pairs' filter kernels (measured at 10k x 10k: 0.072 -> 0.052 ms per pair at depth 3).
    Results of submit() number i live in slot i % depth until submit() number i + depth reuses it."""

    def __init__(self, nq, nt, device, ratio=0.70, depth=3, filter="auto"):
        self.depth = int(depth)
        self.matchers = [PairMatcher(nq, nt, device, ratio, filter=filter) for _ in range(self.depth)]
        self.streams = _pipeline_streams(self.depth, device)
        self.n = 0

    def submit(self, des0, des1, after=None, result=None):
        """Enqueue one pair on the next stream; returns (slot, stream, (idx, dist, out_q, out_t, count)) — the tensors
        are valid once `stream` reaches here, and are overwritten by submit() number i + depth.
        `after`: None → the pair waits for everything already enqueued on the caller's current stream (inputs produced
        there are safe); an Event → it waits for that event only (e.g. "the consumer has read this slot's previous
        result"); False → no wait (inputs must already be complete)."""
        k = self.n % self.depth
        self.n += 1
        st = self.streams[k]
        if after is None:
            st.wait_stream(torch.cuda.current_stream(st.device))
        elif after is not False:
            st.wait_event(after)
        with torch.cuda.stream(st):
            out = self.matchers[k].run(des0, des1, result)
        return k, st, out

    def synchronize(self):
        for st in self.streams:
            st.synchronize()


_PROBE_BUF = {}        # device index -> a 64 Mi-float scratch tensor for streams_overlap()
_REJECTED_STREAMS = []  # streams independent_streams() turned down stay alive: destroying one hands its queue slot to the next creation


def streams_overlap(a, b):
    """Does work on stream `b` run beside work queued EARLIER on stream `a`?  HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in
    creation order; two streams on one queue execute strictly one after the other whatever their priorities.  Nothing in the API
    tells which queue a stream got, so: a long elementwise kernel goes to `a`, then a one-wave kernel to `b` — if that one finishes
    while the long one is still running, the streams are independent.  ~0.3 ms of device time."""
    dev = a.device
    buf = _PROBE_BUF.get(dev.index)
    if buf is None:
        buf = _PROBE_BUF[dev.index] = torch.zeros(1 << 26, dtype=torch.float32, device=dev)
        buf[64:].mul_(1.0); buf[:64].mul_(1.0)        # (first launches load the code objects: not inside a timed probe)
    small = buf[:64]
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for s in (a, b):                                  # a stream's hardware queue is set up at its FIRST use (0.3-5 ms): not inside the probe
        with torch.cuda.stream(s):
            small.mul_(1.0)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(a):
        e0.record(a)
        buf[64:].mul_(1.0)
        buf[64:].mul_(1.0)
        ea.record(a)
    with torch.cuda.stream(b):
        small.mul_(1.0)
        eb.record(b)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(eb) < 0.5 * e0.elapsed_time(ea)


_STREAM_SETS = {}      # (kind, device index, depth) -> streams probed ONCE per process (shared_streams)
_STREAM_SETS_LOCK = __import__("threading").Lock()


def shared_streams(kind, depth, device, probe=True):
    """The `depth` streams pipelines of one kind ("pipeline": launch-set / pair pipelines, "sift": frame pipelines) run on, for this
    device: probed ONCE per process to reach different hardware queues (independent_streams) and then shared by every pipeline of
    that kind and depth.  (every constructor used to probe — a 256 MB scratch tensor and several device-wide drains per
    pipeline, also from a producer thread in the middle of a job, where the wall-clock probe can return either answer.)  Pipelines
    that share a stream set run one after the other on it; an owner that needs concurrency between pipelines passes its own.
    probe=False (a non-owner thread, or work in flight): plain streams, nothing drained, nothing cached."""
    device = torch.device(device)
    depth = int(depth)
    if not probe:
        return [torch.cuda.Stream(device=device) for _ in range(depth)]
    key = (kind, device.index if device.index is not None else torch.cuda.current_device(), depth)
    with _STREAM_SETS_LOCK:
        st = _STREAM_SETS.get(key)
        if st is None:
            st = independent_streams(depth, device) if depth > 1 else [torch.cuda.Stream(device=device)]
            release_probe_scratch(device)
            _STREAM_SETS[key] = st
    return list(st)


def _pipeline_streams(depth, device):
    """The streams of a launch-set / pair pipeline (shared_streams: probed once per device and depth, then reused)."""
    return shared_streams("pipeline", depth, device)


def release_probe_scratch(device):
    """Give streams_overlap()'s 256 MB scratch tensor of this device back to the allocator."""
    _PROBE_BUF.pop(torch.device(device).index, None)


def independent_streams(count, device, priority=0, avoid=(), tries=16):
    """`count` new streams that overlap with each other and with the streams in `avoid` (streams_overlap): candidates that landed
    on a hardware queue already taken are set aside and another one is created, at most `tries` candidates in all (then the rest
    are taken as they come).  One-time set-up, a few milliseconds; which streams a job runs on never changes its results."""
    device = torch.device(device)
    taken, out, made = list(avoid), [], 0
    while len(out) < count:
        s = torch.cuda.Stream(device=device, priority=priority)
        made += 1
        if made > tries or all(streams_overlap(t, s) and streams_overlap(s, t) for t in taken + out):     # (not symmetric when priorities differ)
            out.append(s)
        else:
            _REJECTED_STREAMS.append(s)
            del _REJECTED_STREAMS[:-64]                       # (bounded: the oldest go back to the runtime)
    return out


class BatchPipeline:
    """Independent pairs of one shape, `batch` per launch set (BatchMatcher), launch sets pipelined over `depth` HIP streams.
    submit() queues a pair; the batch is launched on the next stream when it is full (or on flush()).  Results of launch
    set i live in matcher i % depth until launch set i + depth reuses it — or in the caller's `result` blocks."""

    def __init__(self, nq, nt, device, ratio=0.70, depth=3, batch=4, filter="auto", streams=None):
        self.depth, self.batch = int(depth), int(batch)
        self.matchers = [BatchMatcher(nq, nt, device, ratio, batch, filter=filter) for _ in range(self.depth)]
        # `streams`: an owner of several pipelines (sharded.HipMatchEngine: one per pair shape) hands all of them the same set
        self.streams = list(streams) if streams is not None else _pipeline_streams(self.depth, device)
        if len(self.streams) != self.depth:
            raise SfmHipError("BatchPipeline: one stream per launch set in flight")
        self.n = 0
        self._pairs, self._results, self._after = [], [], []

    def probe_ms(self, sets, steps=40, warm=10):
        """Milliseconds per launch set over `steps` launch sets of `sets` (a list of lists of (des0, des1) pairs, `batch` each, rotated):
        what tune_streams() compares.  Drains the pipeline before and after."""
        import time

        def run(n):
            for i in range(n):
                for q, t in sets[i % len(sets)]:
                    self.submit(q, t, after=False)
            self.flush()
            self.synchronize()
        run(warm)
        t0 = time.perf_counter()
        run(steps)
        return (time.perf_counter() - t0) / steps * 1e3

    def tune_streams(self, sets, tries=3, steps=40):
        """Two launch sets in flight are ~4 % faster than three (MI355X, 10k x 10k: 0.176 against 0.185 ms) — when the runtime happens to
        serve their two streams concurrently; about one fresh pair of streams in 24 runs them one after the other instead (0.22 ms,
        the one-stream figure; profiles/r05_knn_pipe_depth.txt).  Nothing in the HIP API pins that choice, so: probe this pipeline's
        streams on the caller's data, replace them by fresh ones up to `tries - 1` times, keep the fastest set.  Set-up work, a few
        milliseconds per try; results are unaffected.  Returns the per-launch-set milliseconds of every try."""
        seen = []
        best, best_streams = None, None
        for _ in range(max(1, tries)):
            ms = self.probe_ms(sets, steps)
            seen.append(ms)
            if best is None or ms < best:
                best, best_streams = ms, self.streams
            if len(seen) < tries:                           # (fresh ones, not the device's shared set: this IS the owner's explicit probe)
                self.streams = independent_streams(self.depth, self.streams[0].device) if self.depth > 1 else [torch.cuda.Stream(device=self.streams[0].device)]
                release_probe_scratch(self.streams[0].device)
        self.streams = best_streams
        return seen

    def submit(self, des0, des1, after=None, result=None):
        """Queue one pair.  `after`: None -> the launch set waits for everything already enqueued on the caller's current
        stream when it is launched; an Event -> it waits for that event (e.g. "the consumer has read this result block");
        False -> no wait.  `result`: optional int32 [2][nq][2] block for this pair's (trainIdx, distance bits) — either every
        pair of a batch has one or none has.  Returns the launch record (slot, stream, matcher, pairs) when this pair
        completed a batch, else None."""
        if self._pairs and (result is not None) != bool(self._results):
            raise SfmHipError("BatchPipeline.submit: either every pair of a batch has a result block or none has")
        self._pairs.append((des0, des1))
        if result is not None:
            self._results.append(result)
        self._after.append(after)
        return self.flush() if len(self._pairs) == self.batch else None

    @property
    def pending(self):
        """Pairs queued for the next launch set (not yet launched)."""
        return len(self._pairs)

    def flush(self):
        """Launch the queued pairs (a partial batch is fine).  Returns (slot, stream, matcher, n_pairs) or None."""
        if not self._pairs:
            return None
        k = self.n % self.depth
        self.n += 1
        st, bm = self.streams[k], self.matchers[k]
        if any(a is None for a in self._after):
            st.wait_stream(torch.cuda.current_stream(st.device))
        waited = set()
        for a in self._after:                             # (the pairs of a batch usually share one event)
            if a is not None and a is not False and id(a) not in waited:
                st.wait_event(a)
                waited.add(id(a))
        with torch.cuda.stream(st):
            n = bm.run(self._pairs, self._results if self._results else None)
        self._pairs, self._results, self._after = [], [], []
        return k, st, bm, n

    def synchronize(self):
        for st in self.streams:
            st.synchronize()


def selftest_mfma_accumulation(kind=0, trials_per_wave=50, device="cuda", bf16=None):
    """Largest error of one MFMA against the exact c + sum a_k b_k, in units of 2^-24 (|c| + sum |a_k b_k|), for seven
    operand regimes (sfm_selftest_mfma_accumulation): the hardware property the KNN certificate's chain term assumes to be
    <= 16.  kind: 0 v_mfma_f32_32x32x16_f16, 1 ..._32x32x16_bf16, 2 v_mfma_f32_32x32x8_bf16 (the accumulator init)."""
    if bf16 is not None:
        kind = 1 if bf16 else 0
    ws = torch.empty(32768 + 512, dtype=torch.uint8, device=device)
    out = (ctypes.c_double * 7)()
    with on_device(ws.device):
        check(_lib.lib().sfm_selftest_mfma_accumulation(int(kind), int(trials_per_wave), out, ptr(ws), ws.numel(), stream_ptr()),
              "sfm_selftest_mfma_accumulation")
    return list(out)


def knn_mfma_selftest_result(device="cuda"):
    """(worst E measured, chain-term scale) of the once-per-device self-test the first 16-bit KNN call runs."""
    worst, scale = ctypes.c_double(0), ctypes.c_float(0)
    with on_device(torch.device(device)):
        check(_lib.lib().sfm_knn_mfma_selftest_result(ctypes.byref(worst), ctypes.byref(scale)), "sfm_knn_mfma_selftest_result")
    return worst.value, scale.value


def profile_enable(on=True):
    """False / True, or an int n > 1: the knn filter kernel is launched n times inside each event pair."""
    check(_lib.lib().sfm_profile_enable(int(on)), "sfm_profile_enable")


def profile_read(slot):
    """(total_ms, launches) of a profiling slot since the last read (synchronises)."""
    ms = ctypes.c_double(0)
    n = ctypes.c_int64(0)
    check(_lib.lib().sfm_profile_read(int(slot), ctypes.byref(ms), ctypes.byref(n)), "sfm_profile_read")
    return ms.value, n.value
