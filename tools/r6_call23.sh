#!/bin/bash
# bin sizes for the scenes whose regions never saturate: 256x128 (4,3) / 128x128 (3,3) / 128x64 (3,2) / 256x64 (4,2) px
T=${TAG:-r6_bs}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log
for rep in 1 2; do for sh in "4,3" "3,3" "3,2" "4,2"; do
  export MGS_BIN_SHIFT=$sh
  python tools/stage_times.py --splats 1030000 --graph --tag train_$sh 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --scene fog --graph --tag fog_$sh 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --scene sparse --graph --tag sparse_$sh 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 400000 --graph --tag tiny_$sh 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
done; done
sort -k1,1 -s gpurun_out/${T}_ab.log | awk '{print $1, $6, $7, $10, $11, "total", $13, $(NF-1)}'
