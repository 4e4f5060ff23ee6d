#!/bin/bash
# A build of libkws_mi355x.so whose fast kernel is compiled for another number of waves per SIMD (KWS_FAST_WPS: registers via __launch_bounds__,
# waves per workgroup and the LDS split via kws_fast_plan.cpp) -> ab_tmp/libkws_<name>.so for tools/ab_rate.py.  profiles/r06_occupancy.md.
# usage: tools/build_wps_variant.sh <name> <wps> [-DMACRO ...]      (the other objects are the product's: run `make -C ei-keyword-spotting_amd/csrc` first)
set -e
NAME=$1; WPS=$2; shift; shift
HERE=$(cd "$(dirname "$0")/.." && pwd)
SRC=$HERE/ei-keyword-spotting_amd/csrc
mkdir -p $HERE/ab_tmp $SRC/.obj_var
# DEV=1 in the environment: the development build's objects and -DKWS_DEV_SWITCHES (KWS_DEV_FAST_WAVES = waves per workgroup at run time)
OBJDIR=$SRC/.obj; DEVFLAG=""
if [ "${DEV:-}" = "1" ]; then OBJDIR=$SRC/.obj_dev; DEVFLAG="-DKWS_DEV_SWITCHES"; fi
FLAGS="$DEVFLAG --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -DKWS_BUILDING_LIBRARY -Wall -Wno-unused-function -DKWS_FAST_WPS=$WPS"
/opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize "$@" -c -o $SRC/.obj_var/kws_fast_$NAME.o $SRC/kws_fast.hip &
/opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c -o $SRC/.obj_var/kws_fast_plan_$NAME.o $SRC/kws_fast_plan.cpp &
wait
OBJS=$(ls $OBJDIR/*.o | grep -v -e "kws_fast.o" -e "kws_fast_plan.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $HERE/ab_tmp/libkws_$NAME.so $OBJS $SRC/.obj_var/kws_fast_$NAME.o $SRC/.obj_var/kws_fast_plan_$NAME.o -ldl
echo "built ab_tmp/libkws_$NAME.so (KWS_FAST_WPS=$WPS $*)"
