#!/bin/bash
# like ab_repeat.sh, at the default three frames in flight (the headline): alternating repetitions of base and csrc/libmgs_<tag>.so
C=vk_gaussian_splatting_amd/csrc
cp $C/libmgs.so /tmp/libmgs_base.so
for rep in 1 2 3; do for tag in base "$@"; do
  if [ "$tag" = base ]; then cp /tmp/libmgs_base.so $C/libmgs.so; else cp $C/libmgs_$tag.so $C/libmgs.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 128 --warmup 16 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', $rep, 'if3 fps', round(d['value'],1), 'single', round(d['value_single_frame'],1), 'err', d['error_flags'])"
done; done
cp /tmp/libmgs_base.so $C/libmgs.so
