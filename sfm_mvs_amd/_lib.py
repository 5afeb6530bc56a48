"""ctypes binding of libsfmhip.so (include/sfm_hip.h).

The HIP library is the ONLY implementation of the hot path: if it is missing or a call
fails this module raises — there is no CPU or PyTorch fallback.

One piece of host-only GLUE exists beside it and is not a fallback of a kernel: `pipeline.common_points` keeps the
reference's NumPy definition (sfm.py:215-239) for callers WITHOUT a GPU (file tooling, the CPU test of the golden vectors
the reference's own function produced); with a GPU present the same function runs `sfm_common_points` on the device, and
the driver (`pipeline.run_sfm`) only ever uses the device operator.  It never touches the oracle.
"""
import ctypes
import os

import torch  # noqa: F401  (loads the ROCm runtime torch was built with before our .so binds to it)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFM_HIP_LIB") or os.path.join(_HERE, "lib", "libsfmhip.so")   # (the override is a dev switch for A/B runs of two builds:
#                                                                                          whichever binary is loaded names itself — build_id())

ABI_VERSION = 3


class SfmHipError(RuntimeError):
    """A libsfmhip.so entry point returned a negative status."""


_c = ctypes
_i64, _i32, _int, _f32, _f64, _sz, _vp = (_c.c_int64, _c.c_int32, _c.c_int, _c.c_float, _c.c_double,
                                          _c.c_size_t, _c.c_void_p)

# name -> (restype, argtypes); mirrors include/sfm_hip.h one to one
SIGNATURES = {
    "sfm_abi_version": (_int, []),
    "sfm_last_error": (_c.c_char_p, []),
    "sfm_build_id": (_c.c_char_p, []),
    "sfm_knn2_l2_f32_ws_bytes": (_sz, [_i64, _i64, _int, _int]),
    "sfm_knn2_l2_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _int, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sfm_ratio_compact_ws_bytes": (_sz, [_i64]),
    "sfm_ratio_compact": (_int, [_vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sfm_match_l2_f32_ws_bytes": (_sz, [_i64, _i64, _int, _int]),
    "sfm_match_l2_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _int, _int, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sfm_match_batch_l2_f32_ws_bytes": (_sz, [_i64, _i64, _int, _int, _int]),
    "sfm_match_batch_l2_f32": (_int, [_int, _vp, _i64, _i64, _vp, _i64, _i64, _int, _int, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sfm_gather_matches": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "sfm_common_points": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sfm_knn_merge_top2": (_int, [_vp, _int, _i64, _vp, _vp, _vp]),
    "sfm_mask_indices_ws_bytes": (_sz, [_i64]),
    "sfm_mask_indices": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _sz, _vp]),
    "sfm_triangulate_dlt": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _int, _vp, _vp]),
    "sfm_triangulate_matches_batch_ws_bytes": (_sz, [_int, _i64]),
    "sfm_triangulate_matches_batch": (_int, [_int, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "sfm_project_residual_ws_bytes": (_sz, [_i64, _i64, _i64]),
    "sfm_project_residual": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _f32,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sfm_ba_dense_sweep_ws_bytes": (_sz, [_i64, _i64]),
    "sfm_ba_dense_sweep": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sfm_ba_schur_ws_bytes": (_sz, [_i64, _i64]),
    "sfm_ba_schur_wt": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "sfm_ba_schur_w": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "sfm_ba_schur_solve_ws_bytes": (_sz, [_i64, _i64]),
    "sfm_ba_schur_solve": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _f64, _int, _f64, _int, _vp, _vp,
                                  _c.POINTER(_i32), _c.POINTER(_i32), _vp, _sz, _vp]),
    "sfm_ba_schur_indexed_ws_bytes": (_sz, [_i64]),
    "sfm_ba_schur_indexed": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _int, _vp, _vp, _vp, _sz, _vp]),
    "sfm_block_inverse": (_int, [_vp, _i64, _int, _vp, _vp]),
    "sfm_block_inverse_checked": (_int, [_vp, _i64, _int, _vp, _vp, _vp]),
    "sfm_block_matvec": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "sfm_norm_l2_ws_bytes": (_sz, []),
    "sfm_norm_l2": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _sz, _vp]),
    "sfm_host_epnp": (_int, [_vp, _vp, _vp, _int, _vp, _vp]),
    "sfm_selftest_mfma_accumulation": (_int, [_int, _int, _vp, _vp, _sz, _vp]),
    "sfm_knn_mfma_selftest_result": (_int, [_c.POINTER(_f64), _c.POINTER(_f32)]),
    "sfm_host_p3p": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "sfm_host_five_point": (_int, [_vp, _vp, _vp, _vp]),
    "sfm_host_decompose_essential": (_int, [_vp, _vp, _vp, _vp]),
    "sfm_host_pnp_dlt_init": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "sfm_host_rodrigues": (_int, [_vp, _int, _vp, _vp]),
    "sfm_project_points_f64": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "sfm_find_essential_mat_ws_bytes": (_sz, [_i64]),
    "sfm_find_essential_mat": (_int, [_vp, _vp, _i64, _vp, _f64, _f64, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sfm_recover_pose_ws_bytes": (_sz, [_i64]),
    "sfm_recover_pose": (_int, [_vp, _vp, _vp, _i64, _vp, _f64, _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sfm_solve_pnp_ransac_ws_bytes": (_sz, [_i64]),
    "sfm_solve_pnp_ransac": (_int, [_vp, _vp, _i64, _vp, _int, _f32, _f64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sfm_bgr2gray_u8": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "sfm_pyrdown_u8": (_int, [_vp, _i64, _i64, _int, _vp, _vp]),
    "sfm_sift_ws_bytes": (_sz, [_i64, _i64, _int, _i64]),
    "sfm_sift_detect_and_compute": (_int, [_vp, _i64, _i64, _i64, _int, _f64, _f64, _f64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sfm_score_essential": (_int, [_vp, _int, _vp, _vp, _i64, _f32, _vp, _vp, _vp]),
    "sfm_recover_pose_score": (_int, [_vp, _int, _vp, _vp, _i64, _f64, _int, _vp, _vp, _vp]),
    "sfm_score_pnp": (_int, [_vp, _int, _vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp]),
    "sfm_profile_enable": (_int, [_int]),
    "sfm_host_sync_count": (_i64, []),
    "sfm_pnp_profile_read": (_int, [_c.POINTER(_f64), _int]),
    "sfm_host_poll_count": (_i64, []),
    "sfm_debug_pnp_sweep_server": (_int, [_int]),
    "sfm_debug_set_trace": (_int, [_vp]),
    "sfm_debug_knn_split_delay": (_int, [_int, _int]),
    "sfm_profile_read": (_int, [_int, _c.POINTER(_f64), _c.POINTER(_i64)]),
}

_lib = None


def lib():
    """Load libsfmhip.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C sfm_mvs_amd/csrc`). "
            "There is no CPU fallback for the hot path.")
    handle = ctypes.CDLL(LIB_PATH)
    # the version first: a stale library must say "rebuild", not fail on the first symbol it lacks
    handle.sfm_abi_version.restype = _int
    handle.sfm_abi_version.argtypes = []
    got = handle.sfm_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI {got} != binding ABI {ABI_VERSION}; rebuild (python -c 'import __graft_entry__ as g; g.build()')")
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} (ABI {got}) does not export {name}: header / library mismatch; rebuild") from e
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def build_id():
    """sfm_build_id() of the LOADED binary: 'knn.hip:<code sha256> assoc.hip:<16 digits> ... sfm_hip.h:<16>' (+ ' dev-build')."""
    return lib().sfm_build_id().decode()


def code_hashes_of_binary():
    """{source file name: code hash} the loaded library was compiled from (scripts/knn_code_hash.py computes the same for a tree)."""
    return dict(tok.split(":", 1) for tok in build_id().split() if ":" in tok)


def knn_code_hash_of_binary():
    """The sha256 of csrc/knn.hip's code the loaded library was compiled from (what profiles/*fuzz*.log and the traffic stamps
    must name)."""
    return build_id().split(":", 1)[1].split()[0]


def check(rc, what):
    if rc != 0:
        msg = lib().sfm_last_error()
        raise SfmHipError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SfmHipError("libsfmhip operates on device (HBM) tensors only; got a CPU tensor — "
                              "there is no CPU fallback")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class on_device:
    """`with torch.cuda.device(dev)` without its cost when `dev` already is the current device (the context manager
    resolves the index through os.environ on every entry: 45 us, 17 % of a 57-camera driver run)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        idx = dev.index if isinstance(dev, torch.device) else dev
        self.ctx = None if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def raw_stream(device_index=None):
    """Handle (int) of the current HIP stream of a device."""
    idx = torch.cuda.current_device() if device_index is None else device_index
    return _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(idx).cuda_stream


def stream_ptr():
    """The current HIP stream of the current device as a void*.  (torch.cuda.current_stream() builds a Stream object and
    resolves the device index through os.environ: 11 us a call, 9 % of a 57-camera driver run; the raw getter is 0.3 us.)"""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
