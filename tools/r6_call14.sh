#!/bin/bash
# frames in flight as a convoy (equal stream priorities) or staggered by stream priority?
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for rep in 1 2; do
for pr in "" "-1,0,1" "-1,0,0" "-1,-1,0" "0,0,1"; do
  MGS_BENCH_STREAM_PRIO="$pr" MGS_BENCH_DUMP_INTERVALS=1 python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-extras 2>/tmp/err.txt | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio [$pr] steps128 |', round(d['value'],1), {k: round(v,3) for k,v in d['frame_interval_ms_percentiles'].items() if k!='note'})"
  grep INTERVALS /tmp/err.txt | cut -c1-160
  MGS_BENCH_STREAM_PRIO="$pr" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio [$pr] steps20  |', round(d['value'],1))"
done; done
