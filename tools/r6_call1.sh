#!/bin/bash
# round 6, GPU call 1: the suite on the new build; same-box A/B of the device-chosen sort partition size (MGS_OS_PART_MIN=4096 =
# the fixed size of rounds 3-5) on garden / train-sized / a middle strip of eight / configs[4]; os_trace of the three new workloads
T=${TAG:-r6_a}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${T}_gpu_tests.log
for rep in 1 2; do
  for pm in 4096 1024; do
    MGS_OS_PART_MIN=$pm python tools/stage_times.py --tag garden_pm$pm 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    MGS_OS_PART_MIN=$pm python tools/stage_times.py --splats 1030000 --tag train_pm$pm 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    MGS_OS_PART_MIN=$pm python tools/stage_times.py --strip 34 38 --tag strip34_38_pm$pm 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    MGS_OS_PART_MIN=$pm python tools/stage_times.py --instances 8 --frames 16 --tag x8_pm$pm 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  done
done
L=$PWD/vk_gaussian_splatting_amd/csrc/libmgs_trace.so
MGS_LIB=$L MGS_GRAPH=0 MGS_OS_TRACE_FILE=/tmp/o.bin OS_INSTANCES=8 timeout 300 python tools/os_trace.py 0 > gpurun_out/${T}_os_trace_x8.log 2>&1
MGS_LIB=$L MGS_GRAPH=0 MGS_OS_TRACE_FILE=/tmp/o.bin OS_STRIP="34 38" timeout 300 python tools/os_trace.py 0 > gpurun_out/${T}_os_trace_strip.log 2>&1
MGS_LIB=$L MGS_GRAPH=0 MGS_OS_TRACE_FILE=/tmp/o.bin OS_SPLATS=1030000 timeout 300 python tools/os_trace.py 0 > gpurun_out/${T}_os_trace_train.log 2>&1
MGS_LIB=$L MGS_GRAPH=0 MGS_OS_TRACE_FILE=/tmp/o.bin MGS_OS_PART_MIN=4096 OS_STRIP="34 38" timeout 300 python tools/os_trace.py 0 > gpurun_out/${T}_os_trace_strip_pm4096.log 2>&1
cat gpurun_out/${T}_ab.log; tail -3 gpurun_out/${T}_gpu_tests.log
