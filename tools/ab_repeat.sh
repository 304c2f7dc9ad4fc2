#!/bin/bash
# noise-aware A/B: three alternating repetitions of bench.py --inflight 1 for csrc/libmgs.so ("base") and csrc/libmgs_<tag>.so
C=vk_gaussian_splatting_amd/csrc
cp $C/libmgs.so /tmp/libmgs_base.so
for rep in 1 2 3; do for tag in base "$@"; do
  if [ "$tag" = base ]; then cp /tmp/libmgs_base.so $C/libmgs.so; else cp $C/libmgs_$tag.so $C/libmgs.so; fi
  python bench.py --no-cpu-baseline --inflight 1 --sh-format 1 --steps 64 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', $rep, round(d['value'],1), {k: round(v, 4) for k, v in d['stage_ms_single_stream'].items()})"
done; done
cp /tmp/libmgs_base.so $C/libmgs.so
