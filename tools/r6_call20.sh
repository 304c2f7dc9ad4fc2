#!/bin/bash
# round 6: which partition size for a small sort (flat look-back)?  MGS_OS_PART_MIN sweep, same box, alternating
T=${TAG:-r6_x}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log
for rep in 1 2; do for v in 4096 3072 2560 2048 1536; do
  export MGS_OS_PART_MIN=$v
  python tools/stage_times.py --strip 34 38 --tag strip_pm$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 1030000 --tag train_pm$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 2800000 --tag mid2p8M_pm$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 400000 --tag tiny400k_pm$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
done; done
sort -k1,1 -s gpurun_out/${T}_ab.log | awk '{print $1, $4, $5, "total", $13}'
