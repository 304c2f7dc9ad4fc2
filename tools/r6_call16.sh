#!/bin/bash
# frames in flight vs the driver-shaped command (20 timed frames): 2 / 3 / 4 / 5 contexts, same box, alternating
for rep in 1 2 3; do for k in 2 3 4 5; do
  python bench.py --steps 20 --warmup 5 --inflight $k --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $k steps 20 |', round(d['value'],1))"
done; done
for k in 2 3 4; do python bench.py --inflight $k --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $k steps 128 |', round(d['value'],1))"; done
