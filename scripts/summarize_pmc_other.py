"""Turns the PMC passes of scripts/collect_pmc_other.sh (gpurun_out/pmc_<tag>_{tri,ba,sift}) into profiles/<tag>_other_pmc.md:
per kernel the mean counters per launch and the derived figures DESIGN.md and docs/*.md quote (HBM bytes, VALU instructions per work item,
issue / wait split)."""
import collections, csv, glob, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(wl):
    agg = collections.defaultdict(lambda: collections.defaultdict(dict))     # kernel -> counter -> dispatch -> value
    for f in sorted(glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}_{wl}", "pass*_counter_collection.csv"))):
        order = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if k.startswith("at::") or k.startswith("__amd") or k.startswith("Cijk"):
                continue
            d = order.setdefault(k, {})
            idx = d.setdefault(r["Dispatch_Id"], len(d))                     # n-th launch of this kernel in this pass
            agg[k][r["Counter_Name"]][idx] = float(r["Counter_Value"])
    return agg


def mean(d, sel=None):
    v = [x for i, x in d.items() if sel is None or i in sel]
    return sum(v) / len(v) if v else float("nan")


def table(title, m, notes):
    print(f"### {title}\n")
    print("| counter | mean per launch |\n|---|---|")
    for c in sorted(m):
        print(f"| {c} | {m[c]:.4g} |")
    print()
    for n in notes:
        print(n)
    print()


print(f"# {tag}: PMC counters of the non-KNN legs (triangulation, dense BA sweep / Schur products, SIFT)\n")
print("Collected with `rocprofv3 --kernel-trace --pmc <group>` in four separate passes per workload (scripts/collect_pmc_other.sh:")
print("`bench.py --workload tri|ba|sift --steps 3 --warmup 1`).  FETCH_SIZE / WRITE_SIZE are in KiB; per MI355X_MICROARCH.md FETCH_SIZE")
print("under-counts wide coalesced reads by 2x on gfx950, so HBM read bytes ~= 2 x FETCH_SIZE x 1024.  SQ_* cycle counters are quad-cycles.\n")

tri = load("tri")
k = "triangulate_kernel<4>"
if k in tri:
    n = 10_000_000
    # launch order of bench.py --workload tri: triangulate_kernel<4> runs warmup 1 + 3 steps faithful, 2 + 3 fast, 1 faithful; the
    # guarded path's first pass is its own kernel since round 4 (triangulate_guarded_kernel)
    groups = {"faithful (normalise_w = 1, OpenCV's Jacobi sweeps)": {0, 1, 2, 3, 9}, "fast (normalise_w = 2, inverse iteration)": {4, 5, 6, 7, 8}}
    print("## triangulation, 10^7 distinct points per launch\n")
    for name, sel in groups.items():
        m = {c: mean(v, sel) for c, v in tri[k].items()}
        notes = []
        if "SQ_INSTS_VALU" in m:
            notes.append(f"VALU instructions per point = {m['SQ_INSTS_VALU'] * 64 / n:.0f} (wave-level count x 64 lanes / points); "
                         f"issue-stall share SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.2f}, "
                         f"waitcnt share SQ_WAIT_ANY / SQ_WAVE_CYCLES = {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.2f}, "
                         f"VALU busy share SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']:.2f}")
        if "FETCH_SIZE" in m:
            notes.append(f"HBM traffic ~= {2 * m['FETCH_SIZE'] * 1024 / 1e6:.0f} MB read + {m.get('WRITE_SIZE', 0) * 1024 / 1e6:.0f} MB written per launch "
                         f"= {(2 * m['FETCH_SIZE'] + m.get('WRITE_SIZE', 0)) * 1024 / n:.1f} B per point (algorithmic: 32)")
        table(f"`{k}` — {name}", m, notes)
    for gk in [x for x in tri if x.startswith("triangulate_guarded_kernel")]:
        m = {c: mean(v) for c, v in tri[gk].items()}
        notes = []
        if "SQ_INSTS_VALU" in m:
            ipp = m["SQ_INSTS_VALU"] * 64 / n
            notes.append(f"VALU instructions per point = {ipp:.0f}; at 4 cycles per fp64 wave-instruction on 1024 SIMDs that is {ipp * n / 64 * 4 / 1024 / 2.4e9 * 1e3:.3f} ms "
                         f"at the 2.4 GHz peak clock (the issue roof of this pass); issue-stall share = {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.2f}, "
                         f"waitcnt share = {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.2f}, VALU busy share = {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']:.2f}")
        if "FETCH_SIZE" in m:
            notes.append(f"HBM traffic ~= {2 * m['FETCH_SIZE'] * 1024 / 1e6:.0f} MB read + {m.get('WRITE_SIZE', 0) * 1024 / 1e6:.0f} MB written per launch "
                         f"= {(2 * m['FETCH_SIZE'] + m.get('WRITE_SIZE', 0)) * 1024 / n:.1f} B per point (algorithmic: 32)")
        table(f"`{gk}` — guarded, first pass (normalise_w = 3: fast path only, a lane walks its points, next point prefetched)", m, notes)
    if "triangulate_fixup_kernel" in tri:
        m = {c: mean(v) for c, v in tri["triangulate_fixup_kernel"].items()}
        table("`triangulate_fixup_kernel` (second pass of the guarded path: scan for marks, compacted Jacobi)", m,
              [f"VALU instructions per point of the launch = {m.get('SQ_INSTS_VALU', 0) * 64 / n:.0f} (the marked fraction x the faithful path's count)"])

ba = load("ba")
if ba:
    print("## BASELINE configs[3]: 500 cameras x 200k points (10^8 observations per sweep / product)\n")
    nobs = 1e8
    for k, what in (("ba_dense_kernel<4>", "dense residual / J^T J sweep (full tiles + the partial one)"), ("schur_wt_kernel<4>", "Schur product u = W^T x"), ("schur_w_kernel<4>", "Schur product w = W v"),
                    ("schur_cg_step_kernel", "camera-side CG iteration (one workgroup)"), ("dense_cam_reduce_kernel", "per-camera fold (tree)"),
                    ("schur_cam_fold_kernel", "per-camera fold of a product (tree)"), ("final_reduce_kernel", "reprojection-error fold (tree)")):
        if k not in ba:
            continue
        m = {c: mean(v) for c, v in ba[k].items()}
        notes = []
        if k in ("ba_dense_kernel<4>", "schur_wt_kernel<4>", "schur_w_kernel<4>") and "SQ_INSTS_VALU" in m:
            notes.append(f"VALU instructions per observation = {m['SQ_INSTS_VALU'] * 64 / nobs:.0f}; VALU busy share = {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']:.2f}, "
                         f"issue-stall share = {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.2f}, waitcnt share = {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.2f}")
        if "FETCH_SIZE" in m and k == "ba_dense_kernel<4>":
            notes.append(f"HBM traffic ~= {2 * m['FETCH_SIZE'] * 1024 / 1e6:.0f} MB read + {m.get('WRITE_SIZE', 0) * 1024 / 1e6:.0f} MB written per sweep "
                         f"= {(2 * m['FETCH_SIZE'] + m.get('WRITE_SIZE', 0)) * 1024 / nobs:.1f} B per observation (algorithmic: 8.2); L2 hit rate "
                         f"{m.get('TCC_HIT_sum', 0) / max(m.get('TCC_HIT_sum', 0) + m.get('TCC_MISS_sum', 0), 1):.2f}")
        table(f"`{k}` — {what}", m, notes)

sift = load("sift")
if sift:
    print("## SIFT detectAndCompute, one 968 x 648 frame per launch set (one frame in flight)\n")
    tot = collections.defaultdict(float)
    for k, v in sift.items():
        for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"):
            if c in v:
                tot[(k.split("<")[0], c)] += sum(v[c].values())
    names = sorted({k for k, _ in tot})
    launches = {k.split("<")[0]: 0 for k in sift}
    for k, v in sift.items():
        launches[k.split("<")[0]] += max(len(x) for x in v.values())
    frames = max(1, launches.get("extrema_kernel", 1))
    print("| kernel (all instantiations) | launches / frame | HBM read MB / frame (2 x FETCH_SIZE) | HBM written MB / frame | VALU wave-instructions / frame | waitcnt share |\n|---|---|---|---|---|---|")
    for k in names:
        rd = 2 * tot.get((k, "FETCH_SIZE"), 0) * 1024 / 1e6 / frames
        wr = tot.get((k, "WRITE_SIZE"), 0) * 1024 / 1e6 / frames
        wc = tot.get((k, "SQ_WAVE_CYCLES"), 0)
        print(f"| `{k}` | {launches[k] / frames:.1f} | {rd:.2f} | {wr:.2f} | {tot.get((k, 'SQ_INSTS_VALU'), 0) / frames:.3g} | {tot.get((k, 'SQ_WAIT_ANY'), 0) / wc if wc else 0:.2f} |")
    print()
