#!/bin/bash
# runs tools/os_trace.py and tools/prj_trace.py with the trace build csrc/libmgs_trace.so swapped in (GPU box copy only)
C=vk_gaussian_splatting_amd/csrc
cp $C/libmgs.so /tmp/base.so; cp $C/libmgs_trace.so $C/libmgs.so
MGS_GRAPH=0 MGS_OS_TRACE_FILE=/tmp/o.bin timeout 300 python tools/os_trace.py ${1:-0} > gpurun_out/${TAG:-r3}_os_trace.log 2>&1
MGS_GRAPH=0 MGS_PRJ_TRACE_FILE=/tmp/p.bin timeout 300 python tools/prj_trace.py ${1:-0} > gpurun_out/${TAG:-r3}_prj_trace.log 2>&1
cp /tmp/base.so $C/libmgs.so
