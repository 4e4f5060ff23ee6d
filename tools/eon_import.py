#!/usr/bin/env python3
"""Model ingestion: Edge Impulse EON export  ->  .kwsm model blob.

Reads the two generated files of an Edge Impulse "EON compiled" export
  <export>/tflite-model/trained_model_compiled.cpp   (tensor table, node table, weights;
        reference layout: MODEL/tflite-model/trained_model_compiled.cpp:70-328)
  <export>/model-parameters/model_metadata.h         (labels, ei_dsp_config_mfcc_t, sizes;
        reference layout: MODEL/model-parameters/model_metadata.h:38-132)
and writes ONE little-endian binary blob that both the HIP library (kws_model_load) and the
test oracle (kwso_model_load) understand.  Only *data* is extracted (tensor shapes, quantisation
parameters, weight bytes, op parameters, labels, DSP settings) -- no reference code.

Blob layout (all little endian, 4-byte aligned):
  char  magic[4] = "KWSM"; u32 version = 1 (MFCC block) | 2 (one more i32 behind pre_cof: bits 0..7 DSP block type, 0 = MFCC, 1 = MFE;
                                                             bit 8 EIDSP_QUANTIZE_FILTERBANK = 1, the SDK's default build option)
  u32 n_tensors, n_nodes, n_labels, input_tensor, output_tensor
  u32 raw_sample_count, sampling_frequency, nn_input_frame_size
  dsp : i32 axes, num_cepstral, num_filters, fft_length, win_size, low_frequency, high_frequency, pre_shift
        f32 frame_length, frame_stride, pre_cof
  labels : n_labels x { u32 len; char[len] padded to 4 }
  tensors: n_tensors x { u32 type (TfLiteType: 1 f32, 2 i32, 9 i8); u32 ndims; i32 dims[ndims];
                         u32 is_const; u32 n_quant; f32 scale[n_quant]; i32 zero_point[n_quant];
                         i32 quantized_dimension; u32 nbytes; u8 data[nbytes if is_const] padded to 4 }
  nodes  : n_nodes x { u32 op (0 RESHAPE,1 CONV_2D,2 ADD,3 MAX_POOL_2D,4 FULLY_CONNECTED,5 SOFTMAX,
                               6 DEPTHWISE_CONV_2D); u32 n_in; i32 in[n_in]; u32 n_out; i32 out[n_out];
                       i32 p[8]; f32 beta }
     p = CONV_2D/DEPTHWISE: padding(1 SAME,2 VALID), stride_w, stride_h, activation, dil_w, dil_h, depth_multiplier, 0
         MAX_POOL_2D      : padding, stride_w, stride_h, filter_w, filter_h, activation, 0, 0
         ADD / FULLY_CONNECTED: activation, 0...
     activation: 0 none, 1 relu, 2 relu_n1_to_1, 3 relu6   (TfLiteFusedActivation)
"""
import argparse
import re
import struct
import sys

import numpy as np

OPS = {"OP_RESHAPE": 0, "OP_CONV_2D": 1, "OP_ADD": 2, "OP_MAX_POOL_2D": 3,
       "OP_FULLY_CONNECTED": 4, "OP_SOFTMAX": 5, "OP_DEPTHWISE_CONV_2D": 6}
TYPES = {"kTfLiteFloat32": 1, "kTfLiteInt32": 2, "kTfLiteInt8": 9}
NP_TYPES = {"int8_t": np.int8, "int32_t": np.int32, "float": np.float32, "uint8_t": np.uint8}
PADDING = {"kTfLitePaddingSame": 1, "kTfLitePaddingValid": 2, "kTfLitePaddingUnknown": 0}
ACT = {"kTfLiteActNone": 0, "kTfLiteActRelu": 1, "kTfLiteActRelu1": 2, "kTfLiteActReluN1To1": 2, "kTfLiteActRelu6": 3}


def _numlist(txt):
    return [t for t in re.split(r"[,\s]+", re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)) if t]


def parse_compiled_model(src):
    arrays = {}
    for m in re.finditer(r"const\s+ALIGN\(\d+\)\s+(\w+)\s+(tensor_data\d+)\[([^\]]*)\]\s*=\s*\{(.*?)\};", src, re.S):
        ctype, name, _, body = m.groups()
        arrays[name] = np.array([float(x) if ctype == "float" else int(x) for x in _numlist(body)],
                                dtype=NP_TYPES[ctype])
    dims = {}
    for m in re.finditer(r"const\s+TfArray<\d+,\s*int>\s+tensor_dimension(\d+)\s*=\s*\{\s*\d+,\s*\{([^}]*)\}\s*\};", src):
        dims[int(m.group(1))] = [int(x) for x in _numlist(m.group(2))]
    scales, zeros, qdim = {}, {}, {}
    for m in re.finditer(r"const\s+TfArray<\d+,\s*float>\s+quant(\d+)_scale\s*=\s*\{\s*\d+,\s*\{([^}]*)\}\s*\};", src):
        scales[int(m.group(1))] = [float(x) for x in _numlist(m.group(2))]
    for m in re.finditer(r"const\s+TfArray<\d+,\s*int>\s+quant(\d+)_zero\s*=\s*\{\s*\d+,\s*\{([^}]*)\}\s*\};", src):
        zeros[int(m.group(1))] = [int(x) for x in _numlist(m.group(2))]
    for m in re.finditer(r"const\s+TfLiteAffineQuantization\s+quant(\d+)\s*=\s*\{[^,]*,[^,]*,\s*(\d+)\s*\};", src):
        qdim[int(m.group(1))] = int(m.group(2))

    tbl = re.search(r"const\s+TensorInfo_t\s+tensorData\[\]\s*=\s*\{(.*?)\};\s*const\s+NodeInfo_t", src, re.S).group(1)
    tensors = []
    row_re = re.compile(r"\{\s*(kTfLite\w+),\s*(kTfLite\w+),\s*([^,]+),\s*\(TfLiteIntArray\*\)&tensor_dimension(\d+),\s*(\d+),\s*\{(kTfLite\w+),")
    for i, m in enumerate(row_re.finditer(tbl)):
        alloc, ttype, data, dim_id, nbytes, qtype = m.groups()
        t = {"type": TYPES[ttype], "dims": dims[int(dim_id)], "nbytes": int(nbytes), "const": alloc == "kTfLiteMmapRo",
             "scale": [], "zero": [], "qdim": 0, "data": b""}
        if qtype == "kTfLiteAffineQuantization":
            t["scale"], t["zero"], t["qdim"] = scales[i], zeros[i], qdim.get(i, 0)
        if t["const"]:
            name = re.search(r"tensor_data\d+", data).group(0)
            raw = arrays[name].tobytes()
            assert len(raw) == t["nbytes"], (name, len(raw), t["nbytes"])
            t["data"] = raw
        tensors.append(t)

    io = {}
    for kind in ("inputs", "outputs"):
        for m in re.finditer(r"const\s+TfArray<\d+,\s*int>\s+%s(\d+)\s*=\s*\{\s*\d+,\s*\{([^}]*)\}\s*\};" % kind, src):
            io[(kind, int(m.group(1)))] = [int(x) for x in _numlist(m.group(2))]
    opdata = {}
    for m in re.finditer(r"const\s+(TfLite\w+Params)\s+opdata(\d+)\s*=\s*\{(.*?)\};", src, re.S):
        opdata[int(m.group(2))] = (m.group(1), [x for x in re.split(r"[,\s{}]+", m.group(3)) if x])
    ntbl = re.search(r"const\s+NodeInfo_t\s+nodeData\[\]\s*=\s*\{(.*?)\};", src, re.S).group(1)
    nodes = []
    for m in re.finditer(r"&inputs(\d+),\s*\(TfLiteIntArray\*\)&outputs(\d+),.*?&opdata(\d+)\)\),\s*(OP_\w+)", ntbl):
        i_in, i_out, i_op, opname = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)
        kind, f = opdata[i_op]
        p, beta = [0] * 8, 0.0
        if kind == "TfLiteConvParams":          # padding, stride_w, stride_h, activation, dil_w, dil_h
            p[0:6] = [PADDING[f[0]], int(f[1]), int(f[2]), ACT[f[3]], int(f[4]), int(f[5])]
        elif kind == "TfLiteDepthwiseConvParams":  # padding, stride_w, stride_h, depth_multiplier, activation, dil_w, dil_h
            p[0:7] = [PADDING[f[0]], int(f[1]), int(f[2]), ACT[f[4]], int(f[5]), int(f[6]), int(f[3])]
        elif kind == "TfLitePoolParams":        # padding, stride_w, stride_h, filter_w, filter_h, activation
            p[0:6] = [PADDING[f[0]], int(f[1]), int(f[2]), int(f[3]), int(f[4]), ACT[f[5]]]
        elif kind in ("TfLiteAddParams", "TfLiteFullyConnectedParams"):
            p[0] = ACT[f[0]]
        elif kind == "TfLiteSoftmaxParams":
            beta = float(f[0])
        nodes.append({"op": OPS[opname], "in": io[("inputs", i_in)], "out": io[("outputs", i_out)], "p": p, "beta": beta})
    m_in = re.search(r"inTensorIndices\[\]\s*=\s*\{\s*(\d+)", src)
    m_out = re.search(r"outTensorIndices\[\]\s*=\s*\{\s*(\d+)", src)
    return tensors, nodes, int(m_in.group(1)), int(m_out.group(1))


def parse_metadata(src, dsp_block="auto"):
    def define(name, cast=int):
        return cast(re.search(r"#define\s+%s\s+([^\s]+)" % name, src).group(1))
    labels = re.findall(r'"([^"]*)"', re.search(r"ei_classifier_inferencing_categories\[\]\s*=\s*\{([^}]*)\}", src).group(1))
    mfcc = re.search(r"ei_dsp_config_mfcc_t\s+ei_dsp_config_\d+\s*=\s*\{(.*?)\};", src, re.S)
    mfe = re.search(r"ei_dsp_config_mfe_t\s+ei_dsp_config_\d+\s*=\s*\{(.*?)\};", src, re.S)
    if dsp_block == "mfe" or (dsp_block == "auto" and mfe and not mfcc):
        # the MFE block of the newer SDK copy (L432 model-parameters/model_metadata.h:103-112: axes, frame_length, frame_stride,
        # num_filters, fft_length, low_frequency, high_frequency, win_size); extract_mfe_features applies no pre-emphasis and its
        # feature matrix is [frames][filters] (classifier/ei_run_dsp.h:369-418)
        v = [x.rstrip("f") for x in _numlist(mfe.group(1))]
        dsp = {"axes": int(v[0]), "num_cepstral": int(v[3]), "frame_length": float(v[1]), "frame_stride": float(v[2]),
               "num_filters": int(v[3]), "fft_length": int(v[4]), "win_size": int(v[7]), "low_frequency": int(v[5]),
               "high_frequency": int(v[6]), "pre_cof": 0.0, "pre_shift": 1, "block": 1}
    else:
        v = [x.rstrip("f") for x in _numlist(mfcc.group(1))]
        dsp = {"axes": int(v[0]), "num_cepstral": int(v[1]), "frame_length": float(v[2]), "frame_stride": float(v[3]),
               "num_filters": int(v[4]), "fft_length": int(v[5]), "win_size": int(v[6]), "low_frequency": int(v[7]),
               "high_frequency": int(v[8]), "pre_cof": float(v[9]), "pre_shift": int(v[10])}
    return {"labels": labels, "dsp": dsp,
            "raw_sample_count": define("EI_CLASSIFIER_RAW_SAMPLE_COUNT"),
            "frequency": define("EI_CLASSIFIER_FREQUENCY"),
            "nn_input_frame_size": define("EI_CLASSIFIER_NN_INPUT_FRAME_SIZE")}


def _pad4(b):
    return b + b"\0" * (-len(b) % 4)


def serialise(tensors, nodes, t_in, t_out, meta):
    d = meta["dsp"]
    # 0: MFCC (extract_mfcc_features), 1: MFE (extract_mfe_features, the newer SDK copy); bit 8: EIDSP_QUANTIZE_FILTERBANK = 1
    block = int(d.get("block", 0)) | (0x100 if d.get("quantize_filterbank") else 0)
    out = [b"KWSM", struct.pack("<I", 2 if block else 1),
           struct.pack("<5I", len(tensors), len(nodes), len(meta["labels"]), t_in, t_out),
           struct.pack("<3I", meta["raw_sample_count"], meta["frequency"], meta["nn_input_frame_size"]),
           struct.pack("<8i3f", d["axes"], d["num_cepstral"], d["num_filters"], d["fft_length"], d["win_size"],
                       d["low_frequency"], d["high_frequency"], d["pre_shift"],
                       d["frame_length"], d["frame_stride"], d["pre_cof"])]
    if block:
        out.append(struct.pack("<i", block))
    for lab in meta["labels"]:
        b = lab.encode()
        out += [struct.pack("<I", len(b)), _pad4(b)]
    for t in tensors:
        out.append(struct.pack("<2I", t["type"], len(t["dims"])))
        out.append(struct.pack("<%di" % len(t["dims"]), *t["dims"]))
        out.append(struct.pack("<2I", int(t["const"]), len(t["scale"])))
        out.append(struct.pack("<%df" % len(t["scale"]), *t["scale"]))
        out.append(struct.pack("<%di" % len(t["zero"]), *t["zero"]))
        out.append(struct.pack("<iI", t["qdim"], t["nbytes"]))
        out.append(_pad4(t["data"]))
    for n in nodes:
        out.append(struct.pack("<2I", n["op"], len(n["in"])))
        out.append(struct.pack("<%di" % len(n["in"]), *n["in"]))
        out.append(struct.pack("<I", len(n["out"])))
        out.append(struct.pack("<%di" % len(n["out"]), *n["out"]))
        out.append(struct.pack("<8if", *n["p"], n["beta"]))
    return b"".join(out)


def parse_blob(blob):
    """Inverse of serialise(): .kwsm bytes -> (tensors, nodes, t_in, t_out, meta)."""
    assert blob[:4] == b"KWSM"
    off = [4]

    def rd(fmt):
        v = struct.unpack_from("<" + fmt, blob, off[0])
        off[0] += struct.calcsize("<" + fmt)
        return v

    (version,) = rd("I")
    assert version in (1, 2)
    nt, nn, nl, t_in, t_out = rd("5I")
    raw, freq, nnin = rd("3I")
    d = rd("8i3f")
    dsp = dict(zip(("axes", "num_cepstral", "num_filters", "fft_length", "win_size", "low_frequency", "high_frequency",
                    "pre_shift", "frame_length", "frame_stride", "pre_cof"), d))
    v = rd("i")[0] if version == 2 else 0
    dsp["block"], dsp["quantize_filterbank"] = v & 0xff, (v >> 8) & 1
    labels = []
    for _ in range(nl):
        (ln,) = rd("I")
        labels.append(blob[off[0]:off[0] + ln].decode())
        off[0] += (ln + 3) & ~3
    tensors = []
    for _ in range(nt):
        ttype, nd = rd("2I")
        dims = list(rd("%di" % nd)) if nd else []
        const, nq = rd("2I")
        scale = list(rd("%df" % nq)) if nq else []
        zero = list(rd("%di" % nq)) if nq else []
        qdim, nbytes = rd("iI")
        data = b""
        if const:
            data = blob[off[0]:off[0] + nbytes]
            off[0] += (nbytes + 3) & ~3
        tensors.append({"type": ttype, "dims": dims, "nbytes": nbytes, "const": bool(const), "scale": scale, "zero": zero,
                        "qdim": qdim, "data": data})
    nodes = []
    for _ in range(nn):
        op, ni = rd("2I")
        ins = list(rd("%di" % ni)) if ni else []
        (no,) = rd("I")
        outs = list(rd("%di" % no)) if no else []
        pb = rd("8if")
        nodes.append({"op": op, "in": ins, "out": outs, "p": list(pb[:8]), "beta": pb[8]})
    meta = {"labels": labels, "dsp": dsp, "raw_sample_count": raw, "frequency": freq, "nn_input_frame_size": nnin}
    return tensors, nodes, t_in, t_out, meta


def import_export(export_dir, dsp_block="auto", quantize_filterbank=False):
    """dsp_block: "auto" (MFE only if the metadata instantiates ei_dsp_config_mfe_t and no MFCC block), "mfcc" or "mfe";
    quantize_filterbank: the application builds the SDK with EIDSP_QUANTIZE_FILTERBANK = 1 (its default, SDK/dsp/config.hpp:75-77; a
    compile-time option of the SDK, not recorded in the export -- the reference's demos build with 0)"""
    with open(f"{export_dir}/tflite-model/trained_model_compiled.cpp") as f:
        tensors, nodes, t_in, t_out = parse_compiled_model(f.read())
    with open(f"{export_dir}/model-parameters/model_metadata.h") as f:
        meta = parse_metadata(f.read(), dsp_block)
    meta["dsp"]["quantize_filterbank"] = 1 if quantize_filterbank else 0
    return serialise(tensors, nodes, t_in, t_out, meta), (tensors, nodes, meta)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("export_dir", help="directory holding tflite-model/ and model-parameters/")
    ap.add_argument("out", help="output .kwsm path")
    ap.add_argument("--dsp-block", choices=("auto", "mfcc", "mfe"), default="auto",
                    help="which DSP block the impulse uses (auto: the one model_metadata.h instantiates)")
    ap.add_argument("--quantize-filterbank", action="store_true",
                    help="the application builds the SDK with EIDSP_QUANTIZE_FILTERBANK=1 (the SDK's default; the reference's demos use 0)")
    a = ap.parse_args()
    blob, (tensors, nodes, meta) = import_export(a.export_dir, a.dsp_block, a.quantize_filterbank)
    with open(a.out, "wb") as f:
        f.write(blob)
    print(f"{a.out}: {len(blob)} bytes, {len(tensors)} tensors, {len(nodes)} nodes, labels={meta['labels']}",
          file=sys.stderr)


if __name__ == "__main__":
    main()
