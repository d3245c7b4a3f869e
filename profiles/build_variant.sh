#!/bin/bash
# usage: build_variant.sh NAME FILE "FLAGS" [OBJECT-IT-REPLACES.hip]
# A library variant with ONE source file rebuilt with extra flags (product build of that file: switches must be compile-time), the rest of the
# objects as built:   bash profiles/build_variant.sh NAME nn_grid.hip "-DICP_COOP_ITEMS=3"  ->  icp_rust_amd/lib/libicp_ab_NAME.so
set -e
cd "$(dirname "$0")/../icp_rust_amd/csrc"
NAME=$1; FILE=$2; FLAGS=$3; AS=${4:-$2}
STEM=${AS%.hip}
O=../lib/obj_ab_$NAME; mkdir -p $O
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include $FLAGS -c $FILE -o $O/$STEM.o
OBJS=$(ls ../lib/obj/*.o | grep -v "/$STEM.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libicp_ab_$NAME.so $OBJS $O/$STEM.o
rm -rf $O
ls -la ../lib/libicp_ab_$NAME.so
