"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (our kernels), mean counter value per launch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "anonymous namespace" not in name or "elementwise_kernel_with_index" in name:
            continue
        short = name.split("(anonymous namespace)::", 1)[-1].split("(float", 1)[0].split("(long", 1)[0]
        short = short.replace("(anonymous namespace)::", "")
        per[(short, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (short, _, c), v in per.items():
        acc[short][c].append(v)
for kname in sorted(acc):
    print(kname)
    for c in sorted(acc[kname]):
        v = acc[kname][c]
        print(f"   {c:34s} mean/launch {sum(v)/len(v):16.1f}   launches {len(v)}")
