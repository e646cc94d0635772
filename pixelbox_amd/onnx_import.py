"""Weights-only importer: `image_similarity.onnx` -> PBXW0001 blob.

PixelBox lets users replace `models/image_similarity.onnx` with their own trained model
(reference README.md:58, src/image_hashes/efficientnet.rs:5,12).  This library is not an ONNX runtime: it
implements the one architecture `resources/train.py:30-46` exports -- torchvision EfficientNet-B0 `features`
-> AdaptiveAvgPool2d(1) -> Flatten -> Linear(1280, D) -> Tanh, opset 11, constant-folded, so every BatchNorm
is already folded into its Conv (`train.py:167-174`).  The importer therefore reads only
  * graph.input[0]'s shape           -> H, W
  * the Conv nodes in DEPENDENCY order (the node list is topologically sorted here, so the file's own node order does
    not matter)                      -> stem, then per MBConv block (expand) / depthwise / se_reduce /
                                        se_expand / project, then the head conv; each with weight + bias
  * the final Gemm (or MatMul + Add) -> Linear(1280, D)
checks every shape against the architecture, and writes the blob layout of pixelbox_amd/weights.py.
Anything else (different architecture, unfolded BatchNorm, missing biases) is rejected with a message.

No `onnx` package is needed (none is installed here): the protobuf wire format is decoded directly
(ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5, .input = 11; NodeProto.input = 1, .op_type = 4,
.attribute = 5; TensorProto.dims = 1, .data_type = 2, .float_data = 4, .name = 8, .raw_data = 9).

    python -m pixelbox_amd.onnx_import image_similarity.onnx image_similarity.pbxw
"""
from __future__ import annotations

import struct
import sys

import numpy as np

from . import weights as W


class OnnxImportError(ValueError):
    pass


# ---- minimal protobuf wire-format reader -------------------------------------------------------------------
def _varint(buf: bytes, pos: int):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf: bytes):
    """Yield (field_number, wire_type, value) for one message; length-delimited values are bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos : pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos : pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos : pos + 4]
            pos += 4
        else:
            raise OnnxImportError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _packed_varints(v: bytes):
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


def _tensor(buf: bytes):
    dims, dtype, name, raw, floats = [], None, "", None, []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_varints(v) if wt == 2 else [v]
        elif fno == 2:
            dtype = v
        elif fno == 8:
            name = v.decode()
        elif fno == 9:
            raw = v
        elif fno == 4:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
    if dtype != 1:  # TensorProto.FLOAT
        return name, None
    if raw is not None:
        arr = np.frombuffer(raw, dtype="<f4")
    else:
        arr = np.asarray(floats, dtype=np.float32)
    n = int(np.prod(dims)) if dims else arr.size
    if arr.size != n:
        raise OnnxImportError(f"initializer {name}: {arr.size} values for shape {dims}")
    return name, arr.reshape(dims).astype(np.float32)


def _node(buf: bytes):
    inputs, outputs, op, attrs = [], [], "", {}
    for fno, wt, v in _fields(buf):
        if fno == 1:
            inputs.append(v.decode())
        elif fno == 2:
            outputs.append(v.decode())
        elif fno == 4:
            op = v.decode()
        elif fno == 5:
            aname, aint = "", None
            for f2, w2, v2 in _fields(v):
                if f2 == 1:
                    aname = v2.decode()
                elif f2 == 3:  # AttributeProto.i
                    aint = v2
            attrs[aname] = aint
    return op, inputs, attrs, outputs


def _topological(nodes, available):
    """Nodes in dependency order (stable: among ready nodes, file order).  ONNX requires a topologically sorted node
    list but exporters and graph editors differ in how they order independent branches; the architecture leaves no
    freedom among its Conv nodes (expand -> depthwise -> se_reduce -> se_expand -> project are data-dependent in that
    order, block after block), so sorting by dependencies makes the Conv sequence independent of the file's node order."""
    have = set(available)
    left = list(nodes)
    out = []
    while left:
        rest = []
        progressed = False
        for nd in left:
            if all((not nm) or nm in have for nm in nd[1]):
                out.append(nd)
                have.update(nd[3])
                progressed = True
            else:
                rest.append(nd)
        if not progressed:
            raise OnnxImportError("graph has a cycle or a node input nothing produces: " + ", ".join(sorted({nm for nd in rest for nm in nd[1] if nm and nm not in have})[:4]))
        left = rest
    return out


def _input_hw(buf: bytes):
    """ValueInfoProto -> (H, W) of a [N, 3, H, W] float input (type=2 -> tensor_type=1 -> shape=2 -> dim=1 -> dim_value=1)."""
    for fno, _, v in _fields(buf):
        if fno == 2:
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    for f3, _, v3 in _fields(v2):
                        if f3 == 2:
                            dims = []
                            for f4, _, v4 in _fields(v3):
                                if f4 == 1:
                                    dv = None
                                    for f5, w5, v5 in _fields(v4):
                                        if f5 == 1 and w5 == 0:
                                            dv = v5
                                    dims.append(dv)
                            if len(dims) == 4:
                                return dims[2], dims[3]
    return None, None


def parse_onnx(data: bytes):
    graph = None
    for fno, _, v in _fields(data):
        if fno == 7:
            graph = v
    if graph is None:
        raise OnnxImportError("not an ONNX ModelProto (no graph)")
    inits, nodes, hw = {}, [], (None, None)
    init_names = set()
    inputs = []
    for fno, _, v in _fields(graph):
        if fno == 5:
            name, arr = _tensor(v)
            init_names.add(name)
            if arr is not None:
                inits[name] = arr
        elif fno == 1:
            nodes.append(_node(v))
        elif fno == 11:
            inputs.append(v)
    input_names = []
    for v in inputs:  # the data input is the graph input that is not an initializer
        name = ""
        for fno, _, x in _fields(v):
            if fno == 1:
                name = x.decode()
        input_names.append(name)
        if name not in init_names and hw == (None, None):
            hw = _input_hw(v)
    nodes = _topological(nodes, init_names | set(input_names))
    return inits, [(op, ins, at) for op, ins, at, _ in nodes], hw


def import_onnx(data: bytes, h: int | None = None, w: int | None = None) -> bytes:
    """ONNX bytes -> PBXW0001 blob.  h, w override the input size recorded in the file (dynamic shapes)."""
    inits, nodes, (fh, fw) = parse_onnx(data)
    h, w = h or fh, w or fw
    if not h or not w:
        raise OnnxImportError("input height/width not recorded in the model: pass h= and w=")
    convs = [(ins, at) for op, ins, at in nodes if op == "Conv"]
    if any(op == "BatchNormalization" for op, _, _ in nodes):
        raise OnnxImportError("model contains unfolded BatchNormalization nodes: export in eval mode with constant folding "
                              "(resources/train.py:172)")
    fc_w = fc_b = None
    for op, ins, at in nodes:
        if op == "Gemm":
            fc_w, fc_b = inits.get(ins[1]), inits.get(ins[2]) if len(ins) > 2 else None
            if fc_w is not None and not at.get("transB", 0):
                fc_w = fc_w.T
    if fc_w is None:  # MatMul + Add form
        mm = [ins for op, ins, _ in nodes if op == "MatMul"]
        add = [ins for op, ins, _ in nodes if op == "Add"]
        if mm and mm[-1][1] in inits:
            fc_w = inits[mm[-1][1]].T
            for ins in add:
                for nm in ins:
                    if nm in inits and inits[nm].ndim == 1 and inits[nm].size == fc_w.shape[0]:
                        fc_b = inits[nm]
    if fc_w is None or fc_b is None or fc_w.ndim != 2 or fc_w.shape[1] != 1280:
        raise OnnxImportError("final Linear(1280, D) with bias not found")
    d = int(fc_w.shape[0])
    specs = W.tensor_specs(d)
    conv_specs = [(n, s) for n, s, _ in specs if n.endswith(".w") and not n.startswith("fc.")]
    if len(convs) != len(conv_specs):
        raise OnnxImportError(f"{len(convs)} Conv nodes, EfficientNet-B0 has {len(conv_specs)}")
    tensors = {}
    for (ins, _), (name, shape) in zip(convs, conv_specs):
        if len(ins) < 3 or ins[1] not in inits or ins[2] not in inits:
            raise OnnxImportError(f"{name}: Conv without constant weight + bias (BatchNorm not folded?)")
        wt, bs = inits[ins[1]], inits[ins[2]]
        if len(shape) == 2:  # 1x1 conv stored [O, I]
            want4 = (shape[0], shape[1], 1, 1)
        elif len(shape) == 3:  # depthwise stored [C, k, k]
            want4 = (shape[0], 1, shape[1], shape[2])
        else:
            want4 = tuple(shape)
        if tuple(wt.shape) != tuple(want4):
            raise OnnxImportError(f"{name}: weight shape {tuple(wt.shape)}, expected {want4}")
        if bs.shape != (shape[0],):
            raise OnnxImportError(f"{name}: bias shape {bs.shape}")
        tensors[name] = wt.reshape(shape)
        tensors[name[:-2] + ".b"] = bs
    tensors["fc.w"], tensors["fc.b"] = fc_w, fc_b
    flat = np.concatenate([np.ascontiguousarray(tensors[n], dtype="<f4").reshape(-1) for n, _, _ in specs])
    if flat.size != W.n_floats(d):
        raise OnnxImportError("internal: float count mismatch")
    hdr = W.MAGIC + struct.pack("<IIIIQ", int(h), int(w), d, len(specs), flat.size)
    return hdr + flat.tobytes()


def main(argv):
    if len(argv) not in (3, 5):
        print("usage: python -m pixelbox_amd.onnx_import model.onnx out.pbxw [H W]", file=sys.stderr)
        return 2
    h, w = (int(argv[3]), int(argv[4])) if len(argv) == 5 else (None, None)
    with open(argv[1], "rb") as f:
        blob = import_onnx(f.read(), h, w)
    with open(argv[2], "wb") as f:
        f.write(blob)
    hh, ww, d, _ = W.parse_blob(blob)
    print(f"wrote {argv[2]}: {hh}x{ww} -> {d}-dim, {len(blob)} bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
