#!/bin/bash
# Compiles ONE instantiation of kws_fast_kernel (device code only, ~20 s instead of 100 s for the whole unit) and prints its registers / scratch / LDS:
# the loop behind the register experiments of profiles/r06_occupancy.md.
# usage: tools/fast_one_form.sh "<4, 5, false, false, false, 0, true>" [-DKWS_FAST_WPS=3 ...]     (template arguments: NZ, DG, PROF, FROM_CEP, NET, QCP, MFE)
set -e
FORM=$1; shift
HERE=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cat > $T/one.hip <<EOF
#define KWS_FAST_NO_LAUNCHERS
#include "$HERE/ei-keyword-spotting_amd/csrc/kws_fast.hip"
template __global__ void kws_fast_kernel$FORM(KwsDspPlan, const KwsFastPlan *, const int16_t *, int, float *, float *, int8_t *, float, int, int *, int *, long long *, const float *,
                                              const KwsNnPlan *, const int *, float *, int);
EOF
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DKWS_BUILDING_LIBRARY -fno-slp-vectorize "$@" -I$HERE/ei-keyword-spotting_amd/csrc -S --cuda-device-only -o $T/one.s $T/one.hip 2>&1 | grep -v "hip-link" || true
grep -E "^\s*; (NumVgprs|ScratchSize|Occupancy|LDSByteSize|VGPRBlocks|NumSgprs):|vgpr_spill_count|\.vgpr_count" $T/one.s | head -12
echo "asm: $T/one.s"
