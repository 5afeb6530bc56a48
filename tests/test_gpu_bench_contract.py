"""bench.py's output contract, as the driver reads it: exactly ONE JSON line on stdout with the metric of BASELINE.json, the
whole-job value, the roofline of the dominant kernel measured live, and (default run) the CPU baseline.  Runs the real script
with a handful of steps and without the extra legs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields(hip):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-extras", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "distances/s" and "distances/sec" in d["metric"] and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "BASELINE configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    # value = units of all ranks / timed seconds: 8 pairs of 10k x 10k per step
    assert d["value"] == pytest.approx(8 * 1e8 / (d["ms_per_step"] * 1e-3), rel=1e-6)
    assert d["value"] > 1e12                                       # (north_star's floor is 1e8; anything below 1e12 means a broken path)
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert k in r, k
    # the headline data (uniform float32) run the i8 MFMA body on 8-bit quantised operands: priced against the dense int8 peak
    assert "QUANTISED" in d["dtype"] and r["bound"] == "mfma" and r["unit"] == "TOP/s" and r["peak"] == 5000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-5) and 0.2 < r["frac"] < 1.0
    assert r["achieved"] == pytest.approx(8 * 1e8 * 256 / (r["avg_launch_ms"] * 1e-3) / 1e12, rel=1e-5)     # algorithmic FLOP / live kernel time
    if "sift_like" in d:                                           # (a leg of the full default run)
        assert d["sift_like"]["roofline"]["peak"] == 5000.0 and 0.2 < d["sift_like"]["roofline"]["frac"] < 1.0


def _strict_loads(line):
    def reject(tok):
        raise ValueError(f"non-finite constant {tok} in the bench line")
    return json.loads(line, parse_constant=reject)


def test_driver_command_line_gives_one_short_strict_json_line(hip):
    """The driver's own command, unmodified (BENCH_rNN.json: `python3 bench.py --gpus 1 --steps 20 --warmup 5`).  Round 5's line was 22.5 KB
    and the driver, which keeps an 8 KB tail of stdout + stderr, could not parse it: the line must be short, strict JSON, alone
    on stdout, and stderr must stay small enough that the line survives in a combined 8 KB tail."""
    full = os.path.join(ROOT, "gpurun_out", "bench_full.json")
    if os.path.exists(full):
        os.remove(full)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, [l[:200] for l in lines]
    line = lines[0]
    assert len(line) < 6000, len(line)
    assert len(line) + len(out.stderr) < 7500, (len(line), len(out.stderr), out.stderr[-1500:])
    d = _strict_loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["value"] == pytest.approx(8 * 1e8 / (d["ms_per_step"] * 1e-3), rel=1e-6) and d["value"] > 1e12
    r, c = d["roofline"], d["cpu_baseline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "frac_step", "traffic", "algorithmic_bytes_per_launch", "kernel", "avg_launch_ms"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4) and 0.2 < r["frac"] < 1.0 and 0.1 < r["frac_step"] <= r["frac"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == "distances/s" and c["value"] > 1e6 and c["cores"] >= 1
    sec = d["config"]["secondary"]                                  # the other legs of the same run: numbers only
    assert all(not isinstance(v, str) or k.endswith(("_scaling", "_error")) for k, v in sec.items()) and sec["sift_frames_per_sec"] > 100 and sec["sfm57_from_pixels_seconds"] < 1.0
    # everything else went to the side file, itself strict JSON
    rec = _strict_loads(open(full).read())
    assert "extra" in rec and rec["value"] == pytest.approx(d["value"], rel=1e-6)
