"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (our kernels), mean counter value per launch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)     # kernel -> dispatch -> duration in ns (under counter collection: kernels run serialised)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "anonymous namespace" not in name or "elementwise_kernel_with_index" in name:
            continue
        short = name.split("(anonymous namespace)::", 1)[-1].split("(float", 1)[0].split("(long", 1)[0]
        short = short.replace("(anonymous namespace)::", "")
        per[(short, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[short][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for (short, _, c), v in per.items():
        acc[short][c].append(v)
N_XCD, N_SE, N_SIMD = 8, 32, 1024       # MI355X: GRBM counts per XCD, SQ_BUSY_CYCLES per shader engine, MFMA busy per SIMD
for kname in sorted(acc):
    print(kname)
    mean = {}
    for c in sorted(acc[kname]):
        v = acc[kname][c]
        mean[c] = sum(v) / len(v)
        print(f"   {c:34s} mean/launch {mean[c]:16.1f}   launches {len(v)}")
    if dur[kname]:
        d = sum(dur[kname].values()) / len(dur[kname])
        print(f"   {'duration under the profiler':34s} mean/launch {d / 1e3:16.1f} us")
        if "GRBM_GUI_ACTIVE" in mean and d > 0:
            print(f"   {'=> clock (GRBM_GUI_ACTIVE / XCDs / t)':34s} {mean['GRBM_GUI_ACTIVE'] / N_XCD / d:16.3f} GHz")
        if "SQ_BUSY_CYCLES" in mean and "SQ_VALU_MFMA_BUSY_CYCLES" in mean and mean["SQ_BUSY_CYCLES"] > 0:
            busy = (mean["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD) / (mean["SQ_BUSY_CYCLES"] / N_SE)
            print(f"   {'=> matrix pipe busy, share of cycles':34s} {busy:16.3f}")
