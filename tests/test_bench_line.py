"""bench.py's stdout line: compact, strict JSON, whatever the full record holds.  CPU only: the committed
round-5 records (22.5 KB for the default run) are pushed through bench.compact_line."""
import glob
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_compact_line_of_every_committed_record_is_short_strict_json(bench):
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_bench_*.json")))
    assert recs
    for path in recs:
        d = json.load(open(path))
        line = bench.compact_line(d)
        assert len(line) < 6000 and "\n" not in line, (path, len(line))
        c = json.loads(line, parse_constant=lambda t: (_ for _ in ()).throw(ValueError(t)))
        for k in ("metric", "value", "unit", "ms_per_step", "config"):
            assert k in c, (path, k)
        assert c["value"] == pytest.approx(d["value"], rel=1e-8)
        if "roofline" in d:
            assert c["roofline"]["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-4)
        if "cpu_baseline" in d:
            assert set(("value", "unit", "cores", "kind", "sample")) <= set(c["cpu_baseline"])


def test_compact_line_survives_nan_and_oversized_blocks(bench):
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    d["roofline"]["traffic"] = float("nan")
    d["cold_value"] = float("inf")
    d["config"]["secondary"] = {f"k{i}": float(i) for i in range(2000)}        # far beyond the budget: the block is dropped, the contract fields stay
    d["config"]["workload"] = "w" * 5000
    line = bench.compact_line(d)
    assert len(line) < 6000
    c = json.loads(line, parse_constant=lambda t: (_ for _ in ()).throw(ValueError(t)))
    assert c["roofline"]["traffic"] is None and "secondary" not in c["config"] and "roofline" in c and "cpu_baseline" in c
