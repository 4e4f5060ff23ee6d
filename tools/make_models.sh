#!/bin/sh
# Regenerates the synthetic model files under models/ (the reference ships no 49x40 or depthwise-separable model: SURVEY 8(c)).
# Heads calibrated (--logit-std): zero-mean class logits with the shipped impulse's spread on standardised features, activation
# ranges from a calibration set -- so the softmax is not saturated and a score error bar means something.
set -e
cd "$(dirname "$0")/.."
python tools/synth_model.py models/cfg2_mfcc40_int8.kwsm --seed 40 --num-filters 40 --ncep 40 --low 300 --high 0 --logit-std 1.5
python tools/dequantize_model.py models/cfg2_mfcc40_int8.kwsm models/cfg2_mfcc40_f32.kwsm
python - <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
from kws_testlib import SYNTH_SPECS, synth_model_blob
open("models/cfg5_dscnn_mfcc40_int8.kwsm", "wb").write(synth_model_blob(**SYNTH_SPECS["cfg5_dscnn"]))
PY
python tools/dequantize_model.py models/cfg5_dscnn_mfcc40_int8.kwsm models/cfg5_dscnn_mfcc40_f32.kwsm
