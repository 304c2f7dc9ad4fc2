#!/bin/bash
# per-workgroup compositor trace for several debug builds (csrc/libmgs_<tag>.so, built with -DMGS_CMP_TRACE)
C=vk_gaussian_splatting_amd/csrc
for tag in "$@"; do
  cp $C/libmgs_$tag.so $C/libmgs.so
  echo "=== $tag"
  MGS_GRAPH=0 MGS_CMP_TRACE_FILE=/tmp/t.bin python tools/cmp_trace.py 0 17 42 2>&1 | grep -v amdgpu | grep "pose\|kernel span\|duration us\|iterations per\|resident\|phase totals\|workgroups with\|list length"
done
