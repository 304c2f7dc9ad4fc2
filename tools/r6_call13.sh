#!/bin/bash
# where does the driver-shaped run lose its 7 %?  warm-up sweep + the intervals between the frame completions of the timed region
for W in 5 8 12 16 24 32 64; do
  python bench.py --steps 20 --warmup $W --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('warmup $W |', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), d.get('frame_interval_ms_percentiles'))"
done
MGS_BENCH_DUMP_INTERVALS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep -E "^INTERVALS" 
MGS_BENCH_DUMP_INTERVALS=1 python bench.py --steps 20 --warmup 64 --no-cpu-baseline --no-extras 2>&1 | grep -E "^INTERVALS"
