"""sha256 of csrc/knn.hip's CODE: comments and whitespace runs removed (string literals keep their text; runs of whitespace collapse everywhere), so that the fuzz
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


if __name__ == "__main__":
    print(knn_code_hash(sys.argv[1] if len(sys.argv) > 1 else None))
