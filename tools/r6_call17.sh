#!/bin/bash
# round 6: the source table of the sort's first pass, two straddled digit-0 values in one batch (default build) vs value after value (libmgs_nofuse.so)
T=${TAG:-r6_u}; mkdir -p gpurun_out; C=$PWD/vk_gaussian_splatting_amd/csrc; rm -f gpurun_out/${T}_ab.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sort or keys" 2>&1 | grep -E "passed|failed" ) | tee gpurun_out/${T}_tests.log
for rep in 1 2 3; do for v in nofuse fuse; do
  unset MGS_LIB; if [ $v = nofuse ]; then export MGS_LIB=$C/libmgs_nofuse.so; fi
  python tools/stage_times.py --graph --tag garden_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --splats 1030000 --graph --tag train_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  python tools/stage_times.py --strip 34 38 --graph --tag strip_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  if [ $rep = 1 ]; then python tools/stage_times.py --instances 8 --frames 16 --graph --tag x8_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log; fi
done; done
unset MGS_LIB
MGS_LIB=$C/libmgs_trace.so MGS_GRAPH=0 MGS_OS_TRACE_FILE=/tmp/o.bin timeout 300 python tools/os_trace.py 0 2>&1 | grep -v amdgpu > gpurun_out/${T}_os_trace_fuse.log
cat gpurun_out/${T}_ab.log; grep -B12 -A8 "pass 1: " gpurun_out/${T}_os_trace_fuse.log | grep -v "^--$"
