#!/bin/bash
for rep in 1 2 3; do for pr in 48 128 256 512; do
  MGS_BENCH_PREROLL=$pr python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('preroll $pr |', round(d['value'],1), 'single', round(d['value_single_frame'],1))"
done; done
