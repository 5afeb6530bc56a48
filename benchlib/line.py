"""The ONE stdout line (compact, strict JSON) and the full record beside it."""
from .common import *  # noqa: F401,F403

COMPACT_MAX = 5500                 # bytes of the ONE stdout line (the driver keeps an 8 KB tail of stdout + stderr: round 5's 22.5 KB line was not parsed)
FULL_JSON = os.path.join(ROOT, "gpurun_out", "bench_full.json")


def _finite(x):
    """Strict JSON: NaN / Infinity become null."""
    if isinstance(x, float):
        return x if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {str(k): _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return _finite(x.item())
    return x


def _sig(x, n=6):
    """Numbers to n significant digits (the compact line only; the full record keeps every digit)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{n}g}")


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3].rstrip() + "..."


def _pick(d, keys, clip=120, digits=9):
    return {k: _clip(_sig(d[k], digits), clip) for k in keys if isinstance(d, dict) and k in d}


def compact_line(out):
    """The ONE stdout line: the contract's fields + `roofline` + `cpu_baseline`, numbers and short labels only.
    Everything else — `extra`, variants, notes — goes to gpurun_out/bench_full.json.  Always < COMPACT_MAX bytes: optional
    blocks are dropped, last first, until it fits."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "data", "dry_run")
    c = {k: _clip(_sig(out[k], 10), 160) for k in top if k in out}
    c["dtype"] = _clip(out.get("dtype"), 200)
    cfg = out.get("config") or {}
    c["config"] = _pick(cfg, ("workload",), clip=330)
    c["config"].update(_pick(cfg, ("nq", "nt", "dim", "pairs_per_step", "images", "descriptors", "pairs", "frame", "working_size", "cameras", "points", "observations",
                                   "backend", "launched_by", "cold_value", "general_float_value", "rccl_ranks"), clip=80))
    c["config"].update(_pick(cfg, ("parallelism",), clip=240))
    for k in ("exchange", "partition"):
        if k in cfg:
            c["config"][k] = cfg[k]
    if isinstance(cfg.get("secondary"), dict):
        c["config"]["secondary"] = {k: (_clip(v, 100) if isinstance(v, str) else _sig(v)) for k, v in cfg["secondary"].items()
                                    if v is not None and (not isinstance(v, str) or k.endswith(("_scaling", "_error")))}   # numbers (+ the scaling kind, an error if any)
    if isinstance(out.get("roofline"), dict):
        c["roofline"] = _pick(out["roofline"], ("bound", "achieved", "peak", "unit", "frac", "frac_step", "frac_of_sustained", "traffic", "algorithmic_bytes_per_launch",
                                                 "algorithmic_bytes", "kernel", "avg_launch_ms", "launches"), clip=100)
    if isinstance(out.get("cpu_baseline"), dict):
        cb = out["cpu_baseline"]
        c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"), clip=40)
        c["cpu_baseline"]["sample"] = _clip(cb.get("sample"), 260)
        c["cpu_baseline"].update(_pick(cb, ("one_thread_distances_per_sec", "torch_cdist_topk_distances_per_sec", "cpu_model"), clip=60))
        c["cpu_baseline"]["opencv"] = _pick(cb["opencv"], ("value", "unit", "kind", "version", "threads")) if isinstance(cb.get("opencv"), dict) else None
    for k in ("exchange", "parity", "job_seconds", "kernels_ms", "cold_value", "cold_ms_per_step"):      # small, optional: dropped first if the line is too long
        if k in out:
            v = out[k]
            c[k] = {kk: _sig(vv) for kk, vv in v.items() if not isinstance(vv, (str, dict, list))} if isinstance(v, dict) else _sig(v)
    c["full_record"] = "gpurun_out/bench_full.json"
    c = _finite(c)
    for drop in ("kernels_ms", "parity", "exchange", "cold_ms_per_step", "cold_value", "job_seconds"):
        if len(json.dumps(c, allow_nan=False)) < COMPACT_MAX:
            break
        c.pop(drop, None)
    if len(json.dumps(c, allow_nan=False)) >= COMPACT_MAX:
        c["config"].pop("secondary", None)
    line = json.dumps(c, allow_nan=False)
    assert len(line) < COMPACT_MAX and "\n" not in line, len(line)
    return line


def emit(out, json_fd):
    """Full record -> gpurun_out/bench_full.json (never stdout / stderr: the driver's 8 KB tail holds both); compact line -> stdout."""
    try:
        os.makedirs(os.path.dirname(FULL_JSON), exist_ok=True)
        with open(FULL_JSON, "w") as f:
            json.dump(_finite(out), f, indent=1, allow_nan=False)
            f.write("\n")
    except OSError as e:
        print(f"[bench] could not write {FULL_JSON}: {e}", file=sys.stderr)
    os.write(json_fd, (compact_line(out) + "\n").encode())


