"""Code hashes of the library's sources (`--build-id`: all of them, what sfm_build_id() returns).  sha256 of csrc/knn.hip's CODE: comments and whitespace runs removed (string literals keep their text; runs of whitespace collapse everywhere), so that the fuzz
logs and PMC traffic stamps under profiles/ stay valid across documentation-only edits of the kernel source and go stale on
any change of a token.  Used by scripts/fuzz_knn.py, scripts/summarize_pmc.py, bench.py and tests/test_gpu_knn.py.
  python scripts/knn_code_hash.py [path]"""
import hashlib
import os
import re
import sys

_TOKEN = re.compile(r'"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'|//[^\n]*|/\*.*?\*/', re.S)


def code_text(src):
    def keep(m):
        t = m.group(0)
        return t if t[0] in "\"'" else " "          # a comment becomes a separator, a literal stays
    return re.sub(r"\s+", " ", _TOKEN.sub(keep, src)).strip()


def knn_code_hash(path=None):
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sfm_mvs_amd", "csrc", "knn.hip")
    return hashlib.sha256(code_text(open(path, encoding="utf-8").read()).encode()).hexdigest()


CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sfm_mvs_amd", "csrc")
HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sfm_hip.h")


def source_files():
    """Every file libsfmhip.so is compiled from: csrc/*.hip, csrc/*.h and the boundary header; knn.hip first."""
    names = sorted(n for n in os.listdir(CSRC) if n.endswith((".hip", ".h")))
    names.sort(key=lambda n: n != "knn.hip")
    return [os.path.join(CSRC, n) for n in names] + [HEADER]


def source_hashes():
    """{file name: code hash}: knn.hip in full (the KNN fuzz logs and PMC stamps name all 64 digits), the others' first 16 digits."""
    out = {}
    for path in source_files():
        name = os.path.basename(path)
        h = knn_code_hash(path)
        out[name] = h if name == "knn.hip" else h[:16]
    return out


def build_id():
    """What sfm_build_id() of a library built from this tree returns (release build): 'knn.hip:<sha256> assoc.hip:<16> ...'."""
    return " ".join(f"{k}:{v}" for k, v in source_hashes().items())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--build-id":
        print(build_id())
    else:
        print(knn_code_hash(sys.argv[1] if len(sys.argv) > 1 else None))
