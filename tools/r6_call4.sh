#!/bin/bash
# round 6, GPU call 4: per-splat strip footprint bound (k_project) + next-tile prefetch of the chunk sums (k_os_pass<3>);
# csrc/libmgs_base.so = the committed tree before both
T=${TAG:-r6_d}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log; R=$PWD; C=$R/vk_gaussian_splatting_amd/csrc
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${T}_gpu_tests.log
for rep in 1 2 3; do
  for v in base new; do
    unset MGS_LIB
    if [ $v = base ]; then export MGS_LIB=$C/libmgs_base.so; fi
    python tools/stage_times.py --strip 34 38 --graph --tag strip34_38_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --strip 0 12 --graph --tag strip0_12_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --instances 8 --frames 16 --tag x8_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    if [ $rep != 3 ]; then python tools/stage_times.py --graph --tag garden_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log; fi
  done
done
unset MGS_LIB
MGS_LIB=$C/libmgs_trace.so MGS_GRAPH=0 MGS_PRJ_TRACE_FILE=/tmp/p.bin STRIP="34 38" timeout 300 python tools/prj_trace.py 0 > gpurun_out/${T}_prj_trace_strip.log 2>&1
MGS_LIB=$C/libmgs_trace.so MGS_GRAPH=0 MGS_OS_TRACE_FILE=/tmp/o.bin OS_INSTANCES=8 timeout 300 python tools/os_trace.py 0 > gpurun_out/${T}_os_trace_x8.log 2>&1
cat gpurun_out/${T}_ab.log; tail -3 gpurun_out/${T}_gpu_tests.log; grep -v amdgpu gpurun_out/${T}_prj_trace_strip.log | head -12; grep -v amdgpu gpurun_out/${T}_os_trace_x8.log | sed -n 1,22p
