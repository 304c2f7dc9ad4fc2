#!/bin/bash
# round 6: fuzz / soak of the build with the flat look-back + the extended sort child
T=${TAG:-r6_r}; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "key_sort_variants or radix_sort" 2>&1 | grep -E "passed|failed" ) | tee gpurun_out/${T}_fuzz.log
( timeout 1200 python tools/fuzz_differential.py 5000 300 2>&1 | grep -v amdgpu | tail -3 ) | tee -a gpurun_out/${T}_fuzz.log
( MGS_RIDE_SPLIT=2 timeout 600 python tools/fuzz_differential.py 6000 60 2>&1 | grep -v amdgpu | tail -2 ) | tee -a gpurun_out/${T}_fuzz.log
( MGS_OS_FLAT=0 timeout 600 python tools/fuzz_differential.py 7000 60 2>&1 | grep -v amdgpu | tail -2 ) | tee -a gpurun_out/${T}_fuzz.log
( timeout 600 python tools/fuzz_stochastic.py 2>&1 | grep -v amdgpu | tail -2 ) | tee -a gpurun_out/${T}_fuzz.log
( timeout 900 python tools/soak.py 2>&1 | grep -v amdgpu | tail -4 ) | tee -a gpurun_out/${T}_fuzz.log
