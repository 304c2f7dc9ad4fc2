#!/bin/bash
# noise-aware A/B over the scenes that stress the compositor: alternating repetitions of bench.py for csrc/libmgs.so ("base")
# and csrc/libmgs_<tag>.so (MGS_LIB): garden at 1 and 3 frames in flight, fog, train-sized, 4K.  REPS=2 by default.
C=vk_gaussian_splatting_amd/csrc
one() { lib=$1; name=$2; shift 2; MGS_LIB=$lib python bench.py --no-cpu-baseline --no-extras "$@" 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), {k: round(v, 4) for k, v in d['stage_ms_single_stream'].items()})"; }
for rep in $(seq 1 ${REPS:-2}); do for tag in base "$@"; do
  if [ "$tag" = base ]; then L=$C/libmgs.so; else L=$C/libmgs_$tag.so; fi
  one $L "$tag garden_if1" --inflight 1
  one $L "$tag garden_if3"
  one $L "$tag fog_if1" --scene fog --steps 48 --warmup 8 --inflight 1
  one $L "$tag fog_if3" --scene fog --steps 48 --warmup 8
  one $L "$tag train_if1" --splats 1030000 --inflight 1
  one $L "$tag train_if3" --splats 1030000
  one $L "$tag 4k_if1" --width 3840 --height 2160 --inflight 1
done; done
