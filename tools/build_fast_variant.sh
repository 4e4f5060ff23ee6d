#!/bin/bash
# A build of libkws_mi355x.so that differs from the product in kws_fast.hip's compile flags only -> ab_tmp/libkws_<name>.so (tools/ab_rate.py).
# usage: tools/build_fast_variant.sh <name> [-DMACRO ...]      (the other objects are the product's: run `make -C ei-keyword-spotting_amd/csrc` first)
set -e
NAME=$1; shift
HERE=$(cd "$(dirname "$0")/.." && pwd)
SRC=$HERE/ei-keyword-spotting_amd/csrc
mkdir -p $HERE/ab_tmp $SRC/.obj_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -DKWS_BUILDING_LIBRARY -Wall -Wno-unused-function -fno-slp-vectorize "$@" \
    -c -o $SRC/.obj_var/kws_fast_$NAME.o $SRC/kws_fast.hip
OBJS=$(ls $SRC/.obj/*.o | grep -v kws_fast.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $HERE/ab_tmp/libkws_$NAME.so $OBJS $SRC/.obj_var/kws_fast_$NAME.o -ldl
echo "built ab_tmp/libkws_$NAME.so ($*)"
