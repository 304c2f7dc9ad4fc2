#!/bin/bash
# rule A (adaptive, ~384 partitions, floor 1536) vs rule B (2048 up to 640 partitions) vs adaptive with floor 2048
T=${TAG:-r6_x3}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log
for rep in 1 2; do for v in 1536 2049 2048 1537; do
  export MGS_OS_PART_MIN=$v
  python tools/stage_times.py --strip 34 38 --graph --tag strip_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --strip 0 12 --graph --tag strip0_12_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --strip 24 29 --graph --tag strip24_29_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 1030000 --graph --tag train_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 1800000 --graph --tag mid1p8M_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 400000 --graph --tag tiny400k_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --scene sparse --graph --tag sparse_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
done; done
sort -k1,1 -s gpurun_out/${T}_ab.log | awk '{print $1, $4, $5, "total", $13, $(NF-1)}'
