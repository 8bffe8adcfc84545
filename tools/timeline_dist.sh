#!/bin/bash
# Durations of the distance kernel launched back to back (tools/run_kernels.py <wl> 10 dist) under rocprofv3.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
out=$ROOT/gpurun_out/tld_$1
rm -rf "$out"
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d "$out" -o t -- python3 "$ROOT/tools/run_kernels.py" $1 10 dist > "$out.log" 2>&1
python3 - "$out" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]
ks = [x for x in t if "kernel_symbol" in x][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
prev = None
for n, s, e in rows[-12:]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"gap {gap:7.1f} us | {(e - s) / 1e3:8.1f} us  {n.split('(anonymous namespace)::')[-1][:50]}", flush=True)
    prev = e
PY
