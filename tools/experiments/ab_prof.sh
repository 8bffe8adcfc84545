#!/bin/bash
# Kernel-only times of variant builds: one rocprofv3 kernel trace per library, filter kernel average printed.
#   bash tools/experiments/ab_prof.sh lib_a.so lib_b.so ...      (paths relative to the repository root)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
    name=$(basename "$lib" .so)
    out=$ROOT/gpurun_out/ab/$name
    rm -rf "$out"
    timeout -k 10 150 rocprofv3 --kernel-trace --stats -d "$out" -o t -- python3 "$ROOT/tools/experiments/filter_variant.py" --child "$ROOT/$lib" > "$ROOT/gpurun_out/ab_$name.log" 2>&1
    python3 - "$out" "$name" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]
ks = [x for x in t if "kernel_symbol" in x][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name"
for n, cnt, avg in c.execute(q):
    if "filter_f16" in n:
        print(f"{sys.argv[2]:28s} filter_f16_kernel {avg:8.1f} us x{cnt}", flush=True)
PY
done
