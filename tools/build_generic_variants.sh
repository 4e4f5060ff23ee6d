#!/bin/bash
# Development builds of the library that differ only in kws_generic.hip's experiment knobs (batch depths of the build for four waves per
# SIMD): ei-keyword-spotting_amd/libkws_var_<tag>.so, linked from the development build's other objects.
# usage: tools/build_generic_variants.sh "<tag> <UB> <FB> <EB>" ...
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/ei-keyword-spotting_amd/csrc
make -s -C "$CS" dev > /dev/null
for v in "$@"; do
  set -- $v
  tag=$1; mkdir -p "$CS/.obj_var"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -DKWS_BUILDING_LIBRARY -Wall -Wno-unused-function -DKWS_DEV_SWITCHES \
      -DKWS_GEN_UB=$2 -DKWS_GEN_FB=$3 -DKWS_GEN_EB=$4 -c -o "$CS/.obj_var/kws_generic_$tag.o" "$CS/kws_generic.hip" \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Function Name: _Z23kws_spectral_lds_kernelILb0ELi4ELi4" | grep "VGPRs:\|Scratch\|VGPRs Spill" | sed "s/.*remark: *//; s/\[-Rpass.*//" | tr '\n' ' '
  echo " <- $tag (UB $2, FB $3, EB $4)"
  objs=$(ls "$CS"/.obj_dev/*.o | grep -v kws_generic.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ei-keyword-spotting_amd/libkws_var_$tag.so" $objs "$CS/.obj_var/kws_generic_$tag.o" -ldl
done
