#!/bin/bash
# round 6, GPU call 8: per-workgroup trace of the binning kernels (count / scan / emit vs the sweep) + kernel stats
T=${TAG:-r6_h}; mkdir -p gpurun_out; C=$PWD/vk_gaussian_splatting_amd/csrc
for v in 0 1; do
  MGS_DB_SWEEP=$v MGS_LIB=$C/libmgs_trace.so MGS_GRAPH=0 MGS_DB_TRACE_FILE=/tmp/d.bin timeout 300 python tools/db_trace.py 0 20 2>&1 | grep -v amdgpu > gpurun_out/${T}_db_trace_sweep$v.log
  MGS_DB_SWEEP=$v TAG=${T}_sweep$v bash tools/kstats.sh 2>&1 | grep -v amdgpu > gpurun_out/${T}_kstats_sweep$v.log
done
cat gpurun_out/${T}_db_trace_sweep*.log gpurun_out/${T}_kstats_sweep*.log
