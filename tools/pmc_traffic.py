"""profiles/pmc_traffic.json from a round's PMC summaries (profiles/<tag>_<workload>_pmc_summary.txt): per kernel and
workload, mean FETCH_SIZE / WRITE_SIZE per launch and the HBM-side bytes they stand for.

    python3 tools/pmc_traffic.py r2 > profiles/pmc_traffic.json

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB; the factor 2 is the gfx950 FETCH_SIZE
correction of MI355X_MICROARCH.md (HBM section).  Infinity-Cache hits are included in FETCH_SIZE."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (shapes only; no GPU call)

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
out = {"_note": "rocprofv3 --pmc passes (tools/profile_round.sh), mean per launch of one module step at each shape; "
                "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, the x2 is the gfx950 FETCH_SIZE correction of "
                "MI355X_MICROARCH.md (HBM section); Infinity-Cache hits are included in FETCH_SIZE"}
names = {"dist_mfma_v2_kernel": r"dist_mfma_v2_kernel<|^DistCfg<\d, 32, false, 2, 0>, false, true", "gather_stats_kernel": r"^gather_stats_kernel",
         "row_sqnorms_kernel": r"^PrepArgs\)|^row_sqnorms_kernel", "ema_embed_kernel": r"^ema_embed_kernel"}
for wl in ("c1", "c2", "c3", "c4"):
    path = os.path.join(root, f"{tag}_{wl}_pmc_summary.txt")
    if not os.path.exists(path):
        continue
    cur, vals = None, {}
    for line in open(path):
        if line.startswith("   "):
            m = re.match(r"\s+(\S+)\s+mean/launch\s+([0-9.]+)", line)
            if m and cur:
                vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
        elif not line.startswith("=="):
            cur = line.strip()
    n, k, d, update, _ = bench.WORKLOADS[wl]
    entry = {}
    for short, pat in names.items():
        hits = [v for kname, v in vals.items() if re.search(pat, kname) and "FETCH_SIZE" in v]
        if not hits:
            continue
        v = max(hits, key=lambda h: h["FETCH_SIZE"])          # the instantiation this shape runs
        e = {"FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v.get("WRITE_SIZE"),
             "traffic_bytes": (2 * v["FETCH_SIZE"] + (v.get("WRITE_SIZE") or 0.0)) * 1024}
        if short == "dist_mfma_v2_kernel":
            e["algorithmic_bytes"] = 4.0 * (n * d + k * d) + 12.0 * n + 4.0 * k
        entry[short] = e
    entry["source"] = f"profiles/{tag}_{wl}_pmc_summary.txt"
    out[wl] = entry
print(json.dumps(out, indent=1))
