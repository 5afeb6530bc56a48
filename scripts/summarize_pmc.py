"""Turns rocprofv3 --pmc CSV passes into the markdown table committed under profiles/."""
import collections
import csv
import glob
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
variant = sys.argv[4] if len(sys.argv) > 4 else ""      # "i8": the SIFT-like run (exact-integer body); its traffic stamp goes to knn_i8_traffic.json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pass*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[k.find("knn_"):].split("(")[0] if "knn_" in k else (k.split("(")[0][-40:])   # keep template arguments
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
what = ("SIFT-like u8 descriptors (SURVEY 8d (ii), 30 % planted twins): the exact-integer i8 body of knn_filter_q4_kernel" if variant == "i8" else
        "config 2 (uniform float32) through filter = noquant: the fp16 single-product body" if variant == "f16" else
        "config 2 (uniform float32), filter = auto: the i8 MFMA body on 8-bit QUANTISED operands + refine_q8_body")
print(f"# {tag}: PMC counters of the KNN step (10k x 10k, {what}, {pairs} pair(s) per launch set), mean per launch\n")
print("Collected with `rocprofv3 --kernel-trace --pmc <group>` in four separate passes (scripts/collect_profiles.sh).")
print("FETCH_SIZE/WRITE_SIZE are in KiB; per MI355X_MICROARCH.md FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950,")
print("so `hbm_read_bytes ~= 2 * FETCH_SIZE * 1024` (WRITE_SIZE uncalibrated).\n")
for k, v in agg.items():
    if "knn_" not in k and "ratio" not in k:
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(f"## {k}\n")
    print("| counter | mean per launch |\n|---|---|")
    for c in sorted(m):
        print(f"| {c} | {m[c]:.4g} |")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("SQ_INSTS_MFMA", 0) > 0:
        print(f"\nMFMA busy cycles / MFMA instruction = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_INSTS_MFMA']:.1f} "
              f"(32 = v_mfma_f32_32x32x16_{{f16,bf16}} and v_mfma_i32_32x32x32_i8, 64 = v_mfma_f32_32x32x2_f32); per-SIMD busy = "
              f"{m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024:.0f} cycles; matrix-pipe busy fraction of CU-busy time = "
              f"{m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CU_CYCLES']):.2f}")
    if "FETCH_SIZE" in m:
        print(f"\nHBM-side read traffic ~= {2 * m['FETCH_SIZE'] * 1024 / 1e6:.1f} MB (2 x FETCH_SIZE), "
              f"write ~= {m.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB per launch")
    if "TCC_HIT_sum" in m:
        print(f"\nL2 hit rate = {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
    print()

# HBM-side bytes per launch of the dominant kernel, stamped with the kernel source it was measured on (bench.py reports
# it only while csrc/knn.hip is unchanged)
import hashlib, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
for k, v in agg.items():
    if k.startswith("knn_filter_q4_kernel") and "FETCH_SIZE" in v:
        m = {c: sum(x) / len(x) for c, x in v.items()}
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out = {"kernel": k, "pairs_per_launch": pairs, "bytes_per_launch": 2 * m["FETCH_SIZE"] * 1024 + m.get("WRITE_SIZE", 0) * 1024,
               "fetch_size_kib": m["FETCH_SIZE"], "write_size_kib": m.get("WRITE_SIZE"),
               "mfma_pipe_busy_frac": (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * m["SQ_BUSY_CU_CYCLES"])) if "SQ_BUSY_CU_CYCLES" in m else None,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 correction; workload 10k x 10k",
               "source": f"profiles/{tag}_knn_i8_pmc.md" if variant == "i8" else f"profiles/{tag}_knn_f16_pmc.md" if variant == "f16" else f"profiles/{tag}_knn_pmc.md",
               "knn_hip_sha256": hashlib.sha256(open(os.path.join(root, "sfm_mvs_amd", "csrc", "knn.hip"), "rb").read()).hexdigest(),
               "knn_hip_code_sha256": __import__("knn_code_hash").knn_code_hash()}      # (collect_profiles.sh runs on the box the tree's own build travelled to;
        #  bench.py reports the stamp only while the LOADED binary's sfm_build_id() names the same hash)
        json.dump(out, open(os.path.join(d, "..", "knn_i8_traffic.json" if variant == "i8" else "knn_f16_traffic.json" if variant == "f16" else "knn_traffic.json"), "w"), indent=1)
        break
