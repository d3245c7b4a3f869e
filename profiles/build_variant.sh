#!/bin/bash
# A library variant for profiles/ab_libs.py: the evaluation kernels rebuilt with extra flags, the rest of the objects as built.
#   bash profiles/build_variant.sh NAME "-DICP_AB_..."   ->  icp_rust_amd/lib/libicp_ab_NAME.so
set -e
cd "$(dirname "$0")/../icp_rust_amd/csrc"
NAME=$1; FLAGS=$2
O=../lib/obj_ab_$NAME; mkdir -p $O
for f in gn_win gn_pull shard gn gn_fast; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include $FLAGS -c $f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libicp_ab_$NAME.so ../lib/obj/nn_brute.o ../lib/obj/nn_grid.o ../lib/obj/nn_tile.o ../lib/obj/qsort.o $O/gn.o $O/gn_fast.o $O/gn_pull.o $O/gn_win.o ../lib/obj/p2plane.o $O/shard.o ../lib/obj/multi.o ../lib/obj/api.o
rm -rf $O
ls -la ../lib/libicp_ab_$NAME.so
