"""rocprofv3 --kernel-trace output, our kernels only, grouped by (kernel, grid size): calls, average / min / max in us.

One kernel symbol serves every problem shape, so the per-symbol --stats table of a run that visits several shapes
(the default bench.py line: headline, per-configuration table, series point) averages over all of them; grouped by
grid size the headline shape stands on its own row (C3: the fused distance + argmin kernel with 512 workgroups).

    python3 tools/trace_by_grid.py <rocprofv3 output dir> > profiles/<tag>_kernels_by_grid.csv"""
import csv
import glob
import re
import sys
from collections import defaultdict

rows = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)      # left mangled by the profiler (a _Float16 parameter)
        if m:
            short = name[m.end():m.end() + int(m.group(1))]
        elif "anonymous namespace" not in name or "at::native" in name:
            continue
        else:
            short = name.split("(anonymous namespace)::", 1)[-1]
            short = short.split("(float", 1)[0].split("(long", 1)[0].split("(int", 1)[0].replace("(anonymous namespace)::", "")
        wg = int(r["Workgroup_Size_X"]) or 1
        rows[(short, int(r["Grid_Size_X"]) // wg, wg)].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
out = csv.writer(sys.stdout)
out.writerow(["kernel", "workgroups", "workgroup_size", "calls", "avg_us", "min_us", "max_us", "total_us"])
for (k, g, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    out.writerow([k, g, wg, len(v), f"{sum(v) / len(v):.1f}", f"{min(v):.1f}", f"{max(v):.1f}", f"{sum(v):.1f}"])
