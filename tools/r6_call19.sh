#!/bin/bash
# round 6: the device-chosen small partitions (MGS_OS_PART_MIN) again, now that the look-back's level 2 is flat
T=${TAG:-r6_w}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log
for rep in 1 2 3; do for v in 4096 2048 1024; do
  export MGS_OS_PART_MIN=$v
  python tools/stage_times.py --strip 34 38 --graph --tag strip_pm$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 1030000 --graph --tag train_pm$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --scene sparse --graph --tag sparse_pm$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
done; done
( MGS_OS_PART_MIN=1024 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sort or keys or strips" 2>&1 | grep -E "passed|failed" ) | tee -a gpurun_out/${T}_ab.log
cat gpurun_out/${T}_ab.log
