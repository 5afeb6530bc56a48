"""sfm_mvs_amd — MI355X-native back-end for the incremental-SfM hot path of
FlagArihant2000/sfm-mvs (sfm.py): descriptor 2-NN + Lowe ratio, DLT triangulation,
reprojection residual / J^T J sweeps.  Host code is Python; all arithmetic of the path runs in
hand-written gfx950 HIP kernels behind the C-ABI of include/sfm_hip.h (libsfmhip.so).
"""
from ._lib import LIB_PATH, SfmHipError, lib  # noqa: F401

__all__ = ["LIB_PATH", "SfmHipError", "lib"]
