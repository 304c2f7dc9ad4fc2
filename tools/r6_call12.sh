#!/bin/bash
# how much the driver-shaped command (--steps 20 --warmup 5) loses to a longer warm-up / a longer timed region, same box
for rep in 1 2 3; do
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 64" "--steps 128 --warmup 16" "--steps 20 --warmup 5 --inflight 1"; do
  python bench.py $args --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args |', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'intervals', d.get('frame_interval_ms'))"
done; done
