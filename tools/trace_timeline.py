"""The last launches of a rocprofv3 --kernel-trace run in time order: per launch its duration and the idle gap before it.

    python3 tools/trace_timeline.py <rocprofv3 output dir> [how many launches, default 14]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
prev = None
for s, e, name in rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 14):]:
    short = name.split("(anonymous namespace)::")[-1].split("(")[0][:60]
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"gap {gap:7.1f} us | {(e - s) / 1e3:8.1f} us  {short}", flush=True)
    prev = e
