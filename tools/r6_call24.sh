#!/bin/bash
# the GPU suite under the round's new knobs (each: the whole -m gpu suite)
mkdir -p gpurun_out; rm -f gpurun_out/r6_knobs_gpu_suite.log
for kn in "MGS_OS_FLAT=0" "MGS_OS_PART_MIN=4096" "MGS_OS_PART_MIN=1024" "MGS_BIN_ADAPT=0"; do
  ( env $kn timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | sed "s/^/$kn: /" ) | tee -a gpurun_out/r6_knobs_gpu_suite.log
done
