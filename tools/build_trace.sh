#!/bin/bash
# trace build of the library (per-workgroup phase stamps in k_project and the sort passes) -> csrc/libmgs_trace.so
# (used by tools/trace_run.sh on the GPU box; objects in /tmp, the normal build is not touched)
set -e
C=$(cd "$(dirname "$0")/../vk_gaussian_splatting_amd/csrc" && pwd)
O=/tmp/mgs_trace_obj; mkdir -p $O
for f in mgs_api k_project k_sort k_osort k_raster k_gut; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMGS_OS_TRACE -DMGS_PRJ_TRACE -DMGS_DB_TRACE -DMGS_CMP_TRACE -c $C/$f.hip -o $O/$f.o &
done
wait
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -c $C/host_model.cpp -o $O/host_model.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmgs_trace.so $O/*.o -lz -lpthread -ldl
ls -la $C/libmgs_trace.so
