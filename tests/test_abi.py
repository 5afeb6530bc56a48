"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/sfm_hip.h declares, and refuses to run on host memory (no CPU fallback)."""
import ctypes
import os
import sys
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sfm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sfm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ["sfm_knn2_l2_f32", "sfm_ratio_compact", "sfm_triangulate_dlt", "sfm_project_residual",
                 "sfm_ba_dense_sweep", "sfm_score_essential", "sfm_score_pnp", "sfm_last_error", "sfm_abi_version"]:
        assert must in syms


def test_library_exports_every_declared_symbol():
    import sfm_mvs_amd
    handle = ctypes.CDLL(sfm_mvs_amd.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(handle, s), f"libsfmhip.so does not export {s}"


def test_binding_table_matches_header():
    from sfm_mvs_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.lib().sfm_abi_version() == 3          # ABI 2: the KNN filter variant is a per-call argument; 3: sweep-server exports, per-file build id


def test_ws_bytes_twins():
    from sfm_mvs_amd import _lib
    L = _lib.lib()
    assert L.sfm_knn2_l2_f32_ws_bytes(10000, 10000, 128, 0) > 10000 * 24
    assert L.sfm_knn2_l2_f32_ws_bytes(10, 10, 64, 0) == 0          # dim != 128 unsupported
    assert L.sfm_knn2_l2_f32_ws_bytes(10, 10, 128, 9) == 0         # unknown filter variant
    assert L.sfm_match_batch_l2_f32_ws_bytes(100, 100, 128, 4, 1) == 0   # the fp32-MFMA variant is single-pair
    for f in range(5):
        assert L.sfm_match_l2_f32_ws_bytes(1000, 1000, 128, f) > 0
    assert L.sfm_project_residual_ws_bytes(1000, 1, 1000) > 0
    assert L.sfm_ba_dense_sweep_ws_bytes(500, 200000) > 0


def test_argument_errors_are_reported_without_a_gpu():
    from sfm_mvs_amd import _lib
    L = _lib.lib()
    rc = L.sfm_knn2_l2_f32(None, 4, 128, None, 4, 128, 64, 0, None, None, None, None, 0, None)
    assert rc == -1 and b"dim must be 128" in L.sfm_last_error()
    rc = L.sfm_triangulate_dlt(None, None, None, None, 4, 1, 4, 5, 0, None, None)
    assert rc == -1 and b"rows must be 4 or 6" in L.sfm_last_error()


def test_build_id_names_the_source_and_release_library_reads_no_tuning_overrides():
    """sfm_build_id() = the hash of the csrc/knn.hip code the binary was compiled from (what the fuzz logs and traffic stamps
    under profiles/ name); a release build carries none of the SFM_KNN_* / SFM_TRI_* tuning switches (dev builds only)."""
    import re
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from knn_code_hash import build_id as tree_build_id, knn_code_hash
    from sfm_mvs_amd import _lib
    bid = _lib.build_id()
    assert bid == tree_build_id() and bid.startswith("knn.hip:" + knn_code_hash() + " "), f"{bid}: stale or dev build of libsfmhip.so"
    assert set(_lib.code_hashes_of_binary()) >= {"knn.hip", "sift.hip", "ransac.hip", "triangulate.hip", "host_solvers.h", "sfm_hip.h"}
    text = subprocess.run(["strings", _lib.LIB_PATH], capture_output=True, text=True).stdout
    envs = set(re.findall(r"^SFM_[A-Z0-9_]+$", text, re.M))
    # SFM_KNN_ASSUME_E: honoured only where it widens the certificate (E >= 8); SFM_PNP_PROF: prints host timings at exit
    assert envs <= {"SFM_KNN_ASSUME_E", "SFM_PNP_PROF"}, f"environment switches in the release library: {sorted(envs)}"


def test_cpu_tensors_are_rejected_loudly():
    from sfm_mvs_amd import SfmHipError, ops
    q = torch.zeros((4, 128))
    with pytest.raises(SfmHipError, match="no CPU fallback"):
        ops.knn2(q, q)
    with pytest.raises(SfmHipError):
        ops.triangulate(np.eye(3, 4), np.eye(3, 4), torch.zeros((2, 5)), torch.zeros((2, 5)))


def test_product_never_imports_the_oracle():
    import ast
    pkg = os.path.join(ROOT, "sfm_mvs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                src = open(path).read()
                for node in ast.walk(ast.parse(src)):
                    names = []
                    if isinstance(node, ast.Import):
                        names = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        names = [node.module or ""]
                    assert not any(n.split(".")[0] == "oracle" for n in names), f"{f} imports the oracle"
                assert "liboracle" not in src and "orc_" not in src, f"{f} loads the oracle library"
            elif f.endswith((".hip", ".h", ".cpp", "Makefile")):
                src = open(path).read()
                assert "liboracle" not in src and "sfm_oracle.h" not in src and "orc_" not in src.replace("orc_knn2_l2_f32)", ""), \
                    f"{f} links the oracle"


def test_oracle_side_never_imports_the_product():
    """The oracle, its cv2-named backend and the golden-vector generator must stand on their own: a golden produced
    through product code would compare the product with itself."""
    import ast
    for rel in ("oracle/oracle.py", "tests/oracle_backend.py", "tests/golden/make_golden.py", "tests/np_solvers.py"):
        tree = ast.parse(open(os.path.join(ROOT, rel)).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n.split(".")[0] in ("sfm_mvs_amd", "sfm-mvs_amd") for n in names), f"{rel} imports the product"
    for rel in ("oracle/sfm_oracle.c", "oracle/solvers_oracle.c", "oracle/sift_oracle.c", "oracle/Makefile"):
        src = open(os.path.join(ROOT, rel)).read()
        assert "sfm_hip.h" not in src and "libsfmhip" not in src and "csrc/" not in src, f"{rel} reaches into the product"
