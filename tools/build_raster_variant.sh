#!/bin/bash
# variant of libmgs.so that differs in k_raster.hip's compile flags only (the other objects are the normal build's):
# tools/build_raster_variant.sh NAME "-DFLAG=.." -> csrc/libmgs_NAME.so.  Use with MGS_LIB=<path>.
set -e
NAME=$1; FLAGS=$2
C=$(cd "$(dirname "$0")/../vk_gaussian_splatting_amd/csrc" && pwd)
make -C $C -j8 >/dev/null
O=/tmp/mgs_rvar_$NAME; mkdir -p $O
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $FLAGS -c $C/k_raster.hip -o $O/k_raster.o 2>&1 | grep -i "error" -A5 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmgs_$NAME.so $C/mgs_api.o $C/k_project.o $C/k_sort.o $C/k_osort.o $O/k_raster.o $C/k_gut.o $C/host_model.o -lz -lpthread -ldl
ls -la $C/libmgs_$NAME.so
