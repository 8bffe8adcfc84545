#!/bin/bash
# Per-kernel averages of a few module steps in one regime:  bash tools/experiments/prof_regime.sh c3 collapsed filter
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export T5VQ_REGIME=$2
out=$ROOT/gpurun_out/reg_$1_$2_$3
rm -rf "$out"
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d "$out" -o t -- python3 "$ROOT/tools/run_kernels.py" $1 10 step $3 > "$out.log" 2>&1
python3 - "$out" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]
ks = [x for x in t if "kernel_symbol" in x][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
for n, cnt, avg in c.execute(q):
    if "at::native" not in n and "at6native" not in n:
        print(f"{avg:9.1f} us x{cnt:4d}  {n[:70]}", flush=True)
PY
