#!/bin/bash
# the driver-shaped command with 0 / 16 / 48 / 128 pre-roll frames in front of the timed region, same box, alternating
for rep in 1 2 3; do for pr in 0 16 48 128; do
  MGS_BENCH_PREROLL=$pr python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('preroll $pr |', round(d['value'],1), 'single', round(d['value_single_frame'],1), d['error_flags'])"
done; done
MGS_BENCH_PREROLL=0 python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps128 preroll 0 |', round(d['value'],1))"
python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps128 preroll 48 |', round(d['value'],1))"
