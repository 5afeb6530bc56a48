"""scripts/knn_code_hash.py: the hash that ties the committed fuzz logs / PMC traffic stamps to csrc/knn.hip ignores comments and
whitespace (documentation-only edits keep the evidence valid) and nothing else."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from knn_code_hash import code_text, knn_code_hash  # noqa: E402


def test_comments_and_whitespace_do_not_count_but_every_token_does(tmp_path):
    src = 'int a = 1; // one\n/* block\n comment */ const char* s = "x // not a comment /* nor */"; char c = \'"\';   int  b ;\n'
    assert code_text(src) == 'int a = 1; const char* s = "x // not a comment /* nor */"; char c = \'"\'; int b ;'
    a, b, c, d = (tmp_path / n for n in ("a.hip", "b.hip", "c.hip", "d.hip"))
    a.write_text(src)
    b.write_text(src.replace("// one", "// another remark, longer").replace("   int  b", "\n\n\tint b"))
    c.write_text(src.replace("a = 1", "a = 2"))
    d.write_text(src.replace("not a comment", "no comment"))             # the text of a string literal counts
    assert knn_code_hash(str(a)) == knn_code_hash(str(b))
    assert knn_code_hash(str(a)) != knn_code_hash(str(c))
    assert knn_code_hash(str(a)) != knn_code_hash(str(d))


def test_kernel_source_hash_is_stable_and_hex():
    h = knn_code_hash()
    assert len(h) == 64 and int(h, 16) >= 0 and h == knn_code_hash()
