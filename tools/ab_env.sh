#!/bin/bash
# same-box A/B of an environment knob: tools/ab_env.sh KNOB "bench args" — three alternating repetitions of KNOB=0 / KNOB=1
K=$1; shift
for rep in 1 2 3; do for v in 0 1; do
  env $K=$v python bench.py --no-cpu-baseline --no-extras --inflight 1 $@ 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$K=$v', $rep, round(d['value'],1), {k: round(v, 4) for k, v in d['stage_ms_single_stream'].items()}, 'err', d['error_flags'], 'psnr', d.get('parity',{}).get('psnr_db_min'))"
done; done
