#!/bin/bash
# round 6: adaptive partition size of small sorts (default) vs the fixed 4 096 (MGS_OS_PART_MIN=4096), same box, alternating
T=${TAG:-r6_x2}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sort or keys or strips" 2>&1 | grep -E "passed|failed" ) | tee gpurun_out/${T}_tests.log
for rep in 1 2 3; do for v in 4096 adaptive; do
  unset MGS_OS_PART_MIN; if [ $v = 4096 ]; then export MGS_OS_PART_MIN=4096; fi
  python tools/stage_times.py --strip 34 38 --graph --tag strip_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --strip 0 12 --graph --tag strip0_12_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 1030000 --graph --tag train_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 2800000 --graph --tag mid2p8M_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 1800000 --graph --tag mid1p8M_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 400000 --graph --tag tiny400k_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --scene sparse --graph --tag sparse_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --graph --tag garden_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
done; done
sort -k1,1 -s gpurun_out/${T}_ab.log | awk '{print $1, $4, $5, "total", $13, $NF-1, $(NF-1)}'
