#!/usr/bin/env python3
"""Make the fp32 twin of an int8 .kwsm model: every tensor becomes float32, constants are de-quantised
(w_f = (q - zero_point[ch]) * scale[ch]; int32 biases * their scale).  The reference ships int8 models only
(MODEL/model-parameters/model_metadata.h:51-58) but its SDK carries the float TFLite-Micro kernels
(TFL/kernels/internal/reference/{conv,add,pooling,fully_connected,softmax}.h) and run_inference()'s float branch
(SDK/classifier/ei_run_classifier.h:436-444, 466-482); BASELINE config 2 ("fp32") is this twin.

usage: dequantize_model.py in.kwsm out.kwsm
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import eon_import  # noqa: E402


def dequantize(blob):
    tensors, nodes, t_in, t_out, meta = eon_import.parse_blob(blob)
    out = []
    for t in tensors:
        t = dict(t)
        if not t["scale"]:                      # shape tensors etc.: unchanged
            out.append(t)
            continue
        n = int(np.prod(t["dims"]))
        if t["const"]:
            q = np.frombuffer(t["data"], np.int8 if t["type"] == 9 else np.int32).astype(np.float64)
            scale = np.float32(t["scale"]).astype(np.float64)
            zero = np.float64(t["zero"])
            if len(scale) > 1:                  # per-channel along quantized_dimension
                shape = [1] * len(t["dims"])
                shape[t["qdim"]] = len(scale)
                q = q.reshape(t["dims"])
                f = (q - zero.reshape(shape)) * scale.reshape(shape)
            else:
                f = (q - zero[0]) * scale[0]
            t["data"] = np.float32(f).tobytes()
        t["type"] = 1
        t["nbytes"] = n * 4
        t["scale"], t["zero"], t["qdim"] = [], [], 0
        out.append(t)
    return eon_import.serialise(out, nodes, t_in, t_out, meta)


if __name__ == "__main__":
    with open(sys.argv[1], "rb") as f:
        b = dequantize(f.read())
    with open(sys.argv[2], "wb") as f:
        f.write(b)
    print("%s: %d bytes" % (sys.argv[2], len(b)), file=sys.stderr)
