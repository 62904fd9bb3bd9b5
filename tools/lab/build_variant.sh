#!/bin/bash
# A/B build of the C-ABI library with extra flags on ONE source file: build/lib_<name>.so (load it with DFOL_LIB=...).
# usage: tools/lab/build_variant.sh <name> <file.hip> <flags...>      e.g. tools/lab/build_variant.sh h2_nopf dfol_pair_h2.hip -DDFOL_H2_PREFETCH=0
set -e
NAME=$1; FILE=$2; shift 2
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/dfol_vqa_amd/csrc
mkdir -p $R/build
make -C $C -j4 > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $C/$FILE -o $R/build/${NAME}_${FILE%.hip}.o
OBJS=$(ls $C/*.o | grep -v "/${FILE%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/build/${NAME}_${FILE%.hip}.o -o $R/build/lib_$NAME.so
echo $R/build/lib_$NAME.so
