#!/bin/bash
# per-kernel rocprofv3 stats of bench.py --inflight 1 -> gpurun_out/${TAG}_inflight1_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ps1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps1 -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras --inflight 1 > /tmp/l3 2>&1
cp $(find /tmp/ps1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG:-r3}_inflight1_kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open('$R/gpurun_out/${TAG:-r3}_inflight1_kernel_stats.csv')):
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(4), '%9.1f'%(float(r['AverageNs'])/1000), '%8.1f'%(float(r['MinNs'])/1000), '%8.1f'%(float(r['MaxNs'])/1000))
PY
