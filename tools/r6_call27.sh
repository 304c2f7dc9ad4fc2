#!/bin/bash
# storage order: 16 / 8 / 4 / 1 size classes x Morton (MGS_SIZE_CLASS_SHIFT 0 / 1 / 2 / 4): strips vs the full frame
T=${TAG:-r6_sc}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log
for rep in 1 2; do for sh in 0 1 2 4; do
  export MGS_SIZE_CLASS_SHIFT=$sh
  python tools/stage_times.py --graph --tag garden_cls$sh 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --strip 34 38 --graph --tag strip_cls$sh 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --strip 0 12 --graph --tag strip0_12_cls$sh 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 1030000 --graph --tag train_cls$sh 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
done; done
sort -k1,1 -s gpurun_out/${T}_ab.log | awk '{print $1, $2, $3, $4, $5, $6, $7, $10, $11, "total", $13, $(NF-1)}'
