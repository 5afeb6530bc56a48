"""The measurement / fuzz scripts run on the GPU box only: keep them at least syntactically alive in the CPU suite, and keep
the logs the docs cite present."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_script_compiles():
    paths = sorted(glob.glob(os.path.join(ROOT, "scripts", "*.py")) +
                   [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + glob.glob(os.path.join(ROOT, "benchlib", "*.py")))
    assert len(paths) > 10
    for p in paths:
        compile(open(p, encoding="utf-8").read(), p, "exec")


def test_profiles_cited_by_the_docs_exist():
    """Every `profiles/<file>` named in DESIGN.md / docs/*.md / README.md is committed (wildcards must match something)."""
    missing = []
    for doc in [os.path.join(ROOT, "DESIGN.md"), os.path.join(ROOT, "README.md")] + sorted(glob.glob(os.path.join(ROOT, "docs", "*.md"))):
        for m in re.finditer(r"profiles/([A-Za-z0-9_.*\-]+)", open(doc, encoding="utf-8").read()):
            name = m.group(1).rstrip(".")
            if not name or name.endswith("_") or name in ("r0", "rNN"):
                continue                                    # prose like `profiles/r05_*` shortened further: nothing to check
            if not glob.glob(os.path.join(ROOT, "profiles", name if "*" in name else name + "*")):
                missing.append(f"{os.path.basename(doc)}: profiles/{name}")
    assert not missing, missing
