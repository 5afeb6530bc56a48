"""Device-tensor operators of the hot path: thin, validating wrappers over libsfmhip.so.

Every function takes/returns torch tensors that already live in HBM and enqueues work on the
current HIP stream; nothing here computes on the host.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import SfmHipError, check, ptr, require_cuda, stream_ptr

_ws_cache = {}


def _workspace(device, nbytes):
    """Grow-only per-device scratch buffer (the C-ABI never allocates)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
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


def knn2(des0, des1, return_stats=False):
    """cv2.BFMatcher().knnMatch(des0, des1, k=2) on device (sfm.py:259-260).

    des0 [nq,128] / des1 [nt,128] float32 CUDA tensors (rows may be strided).
    Returns idx [nq,2] int32 (trainIdx), dist [nq,2] float32 (DMatch.distance).
    """
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
    need = lib.sfm_knn2_l2_f32_ws_bytes(nq, nt, dim)
    if need == 0 and nq > 0:
        raise SfmHipError(f"knn2: unsupported shape nq={nq} nt={nt} dim={dim} (dim must be 128)")
    ws = _workspace(des0.device, need)
    ldq = des0.stride(0) if nq > 1 else dim
    ldt = des1.stride(0) if nt > 1 else dim
    with torch.cuda.device(des0.device):
        check(lib.sfm_knn2_l2_f32(ptr(des0), nq, ldq, ptr(des1), nt, ldt, dim, ptr(idx), ptr(dist), ptr(stats),
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
    with torch.cuda.device(idx.device):
        check(_lib.lib().sfm_ratio_compact(ptr(idx), ptr(dist), nq, float(ratio), ptr(out_q), ptr(out_t), ptr(count),
                                           ptr(mask), stream_ptr()), "sfm_ratio_compact")
    return (out_q, out_t, count, mask) if want_mask else (out_q, out_t, count)


def gather_matches(kp0, kp1, out_q, out_t, count):
    """pts0 = kp0[queryIdx], pts1 = kp1[trainIdx] for the ratio survivors (sfm.py:267-268)."""
    require_cuda(kp0, kp1, out_q, out_t, count)
    kp0 = kp0.contiguous().float()
    kp1 = kp1.contiguous().float()
    cap = out_q.shape[0]
    pts0 = torch.empty((cap, 2), dtype=torch.float32, device=kp0.device)
    pts1 = torch.empty((cap, 2), dtype=torch.float32, device=kp0.device)
    with torch.cuda.device(kp0.device):
        check(_lib.lib().sfm_gather_matches(ptr(kp0), ptr(kp1), ptr(out_q), ptr(out_t), ptr(count), cap, ptr(pts0),
                                            ptr(pts1), stream_ptr()), "sfm_gather_matches")
    return pts0, pts1


def match_pair(des0, des1, ratio=0.70):
    """KNN + ratio for one image pair; returns (query_idx, train_idx, dist1) trimmed to the survivors.

    One host sync (reading the survivor count) — this is the find_features() boundary.
    """
    idx, dist = knn2(des0, des1)
    out_q, out_t, count = ratio_compact(idx, dist, ratio)
    m = int(count.item())
    return out_q[:m], out_t[:m], idx, dist
