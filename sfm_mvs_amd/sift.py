"""Image preprocessing + SIFT on the GPU: the host-side mirror of sfm.py:40 (`cv2.pyrDown`), sfm.py:243-244
(`cv2.cvtColor(..., COLOR_BGR2GRAY)`) and sfm.py:246-252 (`cv2.xfeatures2d.SIFT_create().detectAndCompute`).

Thin plumbing over `sfm_bgr2gray_u8`, `sfm_pyrdown_u8` and `sfm_sift_detect_and_compute` (include/sfm_hip.h): device
tensors in, device tensors out, stream ordered.  No CPU path.
"""
import ctypes

import torch

from . import _lib


def bgr2gray(bgr):
    """(H, W, 3) uint8 device tensor, BGR order -> (H, W) uint8."""
    _lib.require_cuda(bgr)
    if bgr.dtype != torch.uint8 or bgr.dim() != 3 or bgr.shape[2] != 3:
        raise _lib.SfmHipError("bgr2gray expects an (H, W, 3) uint8 tensor")
    bgr = bgr.contiguous()
    h, w, _ = bgr.shape
    out = torch.empty((h, w), dtype=torch.uint8, device=bgr.device)
    with _lib.on_device(bgr.device):
        _lib.check(_lib.lib().sfm_bgr2gray_u8(_lib.ptr(bgr), w, h, 3 * w, _lib.ptr(out), _lib.stream_ptr()), "sfm_bgr2gray_u8")
    return out


def pyrdown(img):
    """(H, W) or (H, W, C) uint8 device tensor -> ((H+1)//2, (W+1)//2[, C]) uint8 (5x5 binomial, reflect-101)."""
    _lib.require_cuda(img)
    if img.dtype != torch.uint8 or img.dim() not in (2, 3):
        raise _lib.SfmHipError("pyrdown expects an (H, W[, C]) uint8 tensor")
    img = img.contiguous()
    h, w = img.shape[:2]
    ch = 1 if img.dim() == 2 else img.shape[2]
    out = torch.empty(((h + 1) // 2, (w + 1) // 2) + tuple(img.shape[2:]), dtype=torch.uint8, device=img.device)
    with _lib.on_device(img.device):
        _lib.check(_lib.lib().sfm_pyrdown_u8(_lib.ptr(img), w, h, ch, _lib.ptr(out), _lib.stream_ptr()), "sfm_pyrdown_u8")
    return out


class Sift:
    """`cv2.SIFT_create(nfeatures=0, nOctaveLayers, contrastThreshold, edgeThreshold, sigma)` for one image size.

    The workspace (the whole scale space, 268 MB for a 968 x 648 frame) and the output buffers are allocated once and
    reused by every `run`; results are views into them, valid until the next `run`.
    """

    def __init__(self, width, height, device, n_octave_layers=3, contrast_threshold=0.04, edge_threshold=10.0, sigma=1.6,
                 max_keypoints=1 << 17):
        self.w, self.h, self.device = int(width), int(height), torch.device(device)
        self.params = (int(n_octave_layers), float(contrast_threshold), float(edge_threshold), float(sigma))
        self.cap = int(max_keypoints)
        n = _lib.lib().sfm_sift_ws_bytes(self.w, self.h, self.params[0], self.cap)
        if n == 0:
            raise _lib.SfmHipError("sfm_sift_ws_bytes rejected the configuration")
        self.ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        self.keypoints = torch.empty((self.cap, 8), dtype=torch.float32, device=self.device)
        self.descriptors = torch.empty((self.cap, 128), dtype=torch.float32, device=self.device)
        self.count = torch.zeros(4, dtype=torch.int32, device=self.device)

    def launch(self, gray, want_descriptors=True):
        """Enqueue detectAndCompute on the current stream; no synchronisation.  Read `count`, `keypoints`, `descriptors` later."""
        _lib.require_cuda(gray)
        if gray.dtype != torch.uint8 or gray.dim() != 2 or tuple(gray.shape) != (self.h, self.w) or gray.stride(1) != 1:
            raise _lib.SfmHipError(f"Sift.launch expects a ({self.h}, {self.w}) uint8 tensor with unit column stride")
        nl, ct, et, sg = self.params
        with _lib.on_device(self.device):
            _lib.check(_lib.lib().sfm_sift_detect_and_compute(
                _lib.ptr(gray), self.w, self.h, gray.stride(0), nl, ct, et, sg, self.cap, _lib.ptr(self.keypoints),
                _lib.ptr(self.descriptors) if want_descriptors else None, _lib.ptr(self.count), _lib.ptr(self.ws),
                ctypes.c_size_t(self.ws.numel()), _lib.stream_ptr()), "sfm_sift_detect_and_compute")

    def check_capacity(self):
        """Read the device counters (synchronises); raise if any internal list overflowed; return the keypoint count."""
        n, raw, cand, extrema = (int(v) for v in self.count.tolist())
        if raw > self.cap or cand > self.cap or extrema > 8 * self.cap:
            raise _lib.SfmHipError(f"SIFT found {max(raw, cand, extrema // 8)} keypoints, more than max_keypoints={self.cap}")
        return n

    def run(self, gray, want_descriptors=True):
        """-> (keypoints (n, 8) f32, descriptors (n, 128) f32 or None), device views.  One device->host read (the count)."""
        self.launch(gray, want_descriptors)
        n = self.check_capacity()
        return self.keypoints[:n], (self.descriptors[:n] if want_descriptors else None)


class SiftPipeline:
    """Independent frames pipelined over `depth` HIP streams (one `Sift` = one scale-space workspace + output set per
    stream).  A frame's launch chain is mostly latency: ~50 dependent blur launches of which the small octaves are one
    workgroup each, a single-workgroup ordered compaction, and one long descriptor kernel that leaves CUs free; frames
    on separate streams fill those gaps.  Results of submit() number i live in engine i % depth until submit() number
    i + depth reuses it."""

    def __init__(self, width, height, device, depth=3, streams=None, **params):
        self.depth = int(depth)
        self.engines = [Sift(width, height, device, **params) for _ in range(self.depth)]
        # `streams`: the caller's choice; by default streams probed to reach different hardware queues (two frames on one queue
        # run one after the other: -2..4 % frames/s for a pair that collides; the probe is a few milliseconds of set-up)
        if streams is None:
            from . import ops
            streams = ops.shared_streams("sift", self.depth, device)      # (probed once per device and depth: the probe drains the device)
        self.streams = list(streams)
        if len(self.streams) != self.depth:
            raise ValueError("SiftPipeline: one stream per frame in flight")
        self.n = 0

    def submit(self, gray, after=None, want_descriptors=True):
        """Enqueue one frame on the next stream; returns (slot, stream, engine).  `after` as in ops.PairPipeline.submit:
        None waits for the caller's current stream, an Event for that event, False for nothing."""
        k = self.n % self.depth
        self.n += 1
        st = self.streams[k]
        if after is None:
            st.wait_stream(torch.cuda.current_stream(st.device))
        elif after is not False:
            st.wait_event(after)
        with torch.cuda.stream(st):
            self.engines[k].launch(gray, want_descriptors)
        return k, st, self.engines[k]

    def synchronize(self):
        for st in self.streams:
            st.synchronize()
