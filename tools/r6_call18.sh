#!/bin/bash
# round 6: k_dbin_count requests ids + codes before the frame's count is known (default build) vs after (libmgs_base.so = HEAD)
T=${TAG:-r6_v}; mkdir -p gpurun_out; C=$PWD/vk_gaussian_splatting_amd/csrc; rm -f gpurun_out/${T}_ab.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binning or variants or strips or knobs" 2>&1 | grep -E "passed|failed" ) | tee gpurun_out/${T}_tests.log
for rep in 1 2 3; do for v in base new; do
  unset MGS_LIB; if [ $v = base ]; then export MGS_LIB=$C/libmgs_base.so; fi
  python tools/stage_times.py --graph --tag garden_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 1030000 --graph --tag train_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --width 3840 --height 2160 --graph --tag 4k_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
done; done
unset MGS_LIB
cat gpurun_out/${T}_ab.log
