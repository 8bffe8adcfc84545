#!/bin/bash
# rocprofv3 kernel-trace statistics of any python tool:  bash tools/prof_cmd.sh <tag> <script.py> [args...]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$ROOT/gpurun_out/prof_$tag
rm -rf "$out"
script=$1; shift
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$out" -o t -- python3 "$ROOT/$script" "$@" > "$out.log" 2>&1
python3 - "$out" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]
ks = [x for x in t if "kernel_symbol" in x][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc"
print("avg_us,calls,total_us,min_us,kernel")
for n, cnt, avg, tot, mn in list(c.execute(q))[:25]:
    print(f"{avg:.1f},{cnt},{tot:.0f},{mn:.1f},\"{n[:150]}\"", flush=True)
PY
