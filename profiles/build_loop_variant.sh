#!/bin/bash
# A library variant with gn_loop.hip rebuilt with extra flags (in-kernel phase stamps: -DICP_LOOP_PROFILE), the rest of
# the objects as built:   bash profiles/build_loop_variant.sh NAME "-DICP_LOOP_PROFILE"  ->  icp_rust_amd/lib/libicp_ab_NAME.so
set -e
cd "$(dirname "$0")/../icp_rust_amd/csrc"
NAME=$1; FLAGS=$2
O=../lib/obj_ab_$NAME; mkdir -p $O
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include $FLAGS -c gn_loop.hip -o $O/gn_loop.o
OBJS=$(ls ../lib/obj/*.o | grep -v gn_loop.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libicp_ab_$NAME.so $OBJS $O/gn_loop.o
rm -rf $O
ls -la ../lib/libicp_ab_$NAME.so
