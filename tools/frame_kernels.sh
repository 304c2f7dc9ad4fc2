#!/bin/bash
# per-kernel averages of the benchmark frame (one frame in flight): rocprofv3 --kernel-trace --stats of bench.py
# usage (on the GPU box): tools/frame_kernels.sh <out.csv> [extra bench.py args]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ps1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps1 -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --inflight 1 "$@" > /tmp/l3 2>&1
cp $(find /tmp/ps1 -name "*kernel_stats.csv" | head -1) $R/$OUT
python - <<PY
import csv
tot = 0
for r in csv.DictReader(open("$R/$OUT")):
    if int(r["Calls"]) >= 20:
        print(r["Name"].split("(")[0][:44].ljust(46), r["Calls"], "avg %.1f us" % (float(r["AverageNs"])/1e3), "min %.1f max %.1f" % (float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
