#!/bin/bash
# variant build of the library for A/B experiments: tools/build_variant.sh NAME "-DFLAG=.. -DFLAG2" [files to rebuild ...]
# -> csrc/libmgs_NAME.so (objects in /tmp/mgs_var_NAME; the normal build is not touched).  Use with MGS_LIB=<path>.
set -e
NAME=$1; FLAGS=$2; shift 2
C=$(cd "$(dirname "$0")/../vk_gaussian_splatting_amd/csrc" && pwd)
O=/tmp/mgs_var_$NAME; mkdir -p $O
for f in mgs_api k_project k_sort k_osort k_raster k_gut; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $FLAGS -c $C/$f.hip -o $O/$f.o &
done
wait
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -c $C/host_model.cpp -o $O/host_model.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libmgs_$NAME.so $O/*.o -lz -lpthread -ldl
ls -la $C/libmgs_$NAME.so
