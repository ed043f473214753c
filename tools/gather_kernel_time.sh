#!/bin/bash
# usage: gather_kernel_time.sh LIB_OR_EMPTY TAG  -> prints gather_tile kernel durations (ns) from a kernel trace
cd /tmp && export TMPDIR=/tmp
[ -n "$1" ] && export CCNN_LIB_PATH=$GRAFT_REPO_ROOT/$1
rm -rf /tmp/gk_$2
timeout -k 10 120 rocprofv3 --kernel-trace -d /tmp/gk_$2 -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_gather.py pavia_u f32 20 > /dev/null 2>&1 || exit 1
python3 - "$2" <<'PY'
import sqlite3, sys, glob
db = glob.glob(f"/tmp/gk_{sys.argv[1]}/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select grid_x, count(*), avg(end-start), min(end-start) from kernels where name like '%gather_tile%' group by grid_x order by 1").fetchall()
print(sys.argv[1], [(g, n, round(a), m) for g, n, a, m in rows])
PY
