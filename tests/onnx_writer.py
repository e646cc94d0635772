"""Minimal ONNX (protobuf wire format) WRITER for tests: emits a ModelProto shaped like what
`torch.onnx.export(model, ..., opset_version=11, do_constant_folding=True)` (resources/train.py:172) produces for
the PixelBox embedder -- Conv nodes with folded-BN weight+bias initialisers in graph order, interleaved
Sigmoid/Mul/GlobalAveragePool/ReduceMean nodes, Flatten, Gemm(transB=1), Tanh."""
import struct

import numpy as np

from pixelbox_amd import weights as W


def _varint(x: int) -> bytes:
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _vi(field: int, value: int) -> bytes:
    return _varint(field << 3) + _varint(value)


def tensor(name: str, arr: np.ndarray, raw: bool = True) -> bytes:
    arr = np.ascontiguousarray(arr, dtype="<f4")
    out = b"".join(_vi(1, int(d)) for d in arr.shape) + _vi(2, 1)
    if raw:
        out += _ld(8, name.encode()) + _ld(9, arr.tobytes())
    else:
        out += _ld(4, arr.tobytes()) + _ld(8, name.encode())  # packed float_data
    return out


def node(op: str, inputs, outputs, attrs=None) -> bytes:
    out = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs) + _ld(4, op.encode())
    for k, v in (attrs or {}).items():
        out += _ld(5, _ld(1, k.encode()) + _vi(3, int(v)) + _vi(20, 2))
    return out


def value_info(name: str, dims) -> bytes:
    shape = b""
    for d in dims:
        shape += _ld(1, _ld(2, d.encode()) if isinstance(d, str) else _vi(1, int(d)))
    ttype = _vi(1, 1) + _ld(2, shape)
    return _ld(1, name.encode()) + _ld(2, _ld(1, ttype))


def export_like_torch(blob: bytes, raw: bool = True, gemm: bool = True, with_bn: bool = False, drop_conv: bool = False,
                      node_order=None, inits_first: bool = False) -> bytes:
    """node_order: None (export order), "reversed", or an int seed for a random permutation of the node list -- a graph
    editor / another exporter may list independent branches (and, in a sloppy file, everything) in another order; the
    importer sorts by dependencies.  inits_first: initialisers before the nodes in the byte stream."""
    h, w, d, t = W.parse_blob(blob)
    nodes, inits = [], []
    cur, n = "input", 0

    def conv(prefix, x, shape4):
        nonlocal n
        wn, bn = f"onnx::Conv_{900 + 2 * n}", f"onnx::Conv_{901 + 2 * n}"
        inits.append(tensor(wn, t[prefix + ".w"].reshape(shape4), raw))
        inits.append(tensor(bn, t[prefix + ".b"], raw))
        y = f"conv{n}"
        nodes.append(node("Conv", [x, wn, bn], [y], {"group": 1}))
        n += 1
        return y

    def silu(x):
        nonlocal n
        nodes.append(node("Sigmoid", [x], [x + "_s"]))
        nodes.append(node("Mul", [x, x + "_s"], [x + "_m"]))
        return x + "_m"

    cur = silu(conv("stem", cur, (32, 3, 3, 3)))
    for i, b in enumerate(W.blocks()):
        p = f"b{i}"
        y = cur
        if b.has_expand:
            y = silu(conv(p + ".expand", y, (b.expanded, b.cin, 1, 1)))
        if not (drop_conv and i == 3):
            y = silu(conv(p + ".dw", y, (b.expanded, 1, b.kernel, b.kernel)))
        nodes.append(node("GlobalAveragePool", [y], [y + "_gap"]))
        s = silu(conv(p + ".se_reduce", y + "_gap", (b.squeeze, b.expanded, 1, 1)))
        s = conv(p + ".se_expand", s, (b.expanded, b.squeeze, 1, 1))
        nodes.append(node("Sigmoid", [s], [s + "_g"]))
        nodes.append(node("Mul", [y, s + "_g"], [y + "_se"]))
        y = conv(p + ".project", y + "_se", (b.cout, b.expanded, 1, 1))
        if b.residual:
            nodes.append(node("Add", [cur, y], [y + "_r"]))
            y = y + "_r"
        cur = y
    if with_bn:
        nodes.append(node("BatchNormalization", [cur], [cur + "_bn"]))
    cur = silu(conv("head", cur, (1280, 320, 1, 1)))
    nodes.append(node("GlobalAveragePool", [cur], ["pool"]))
    nodes.append(node("Flatten", ["pool"], ["flat"], {"axis": 1}))
    if gemm:
        inits.append(tensor("4.weight", t["fc.w"], raw))
        inits.append(tensor("4.bias", t["fc.b"], raw))
        nodes.append(node("Gemm", ["flat", "4.weight", "4.bias"], ["lin"], {"alpha": 1, "transB": 1}))
    else:
        inits.append(tensor("fc_wT", t["fc.w"].T.copy(), raw))
        inits.append(tensor("fc_b", t["fc.b"], raw))
        nodes.append(node("MatMul", ["flat", "fc_wT"], ["mm"]))
        nodes.append(node("Add", ["mm", "fc_b"], ["lin"]))
    nodes.append(node("Tanh", ["lin"], ["output"]))
    if node_order == "reversed":
        nodes = nodes[::-1]
    elif node_order is not None:
        nodes = [nodes[i] for i in np.random.default_rng(int(node_order)).permutation(len(nodes))]
    if inits_first:
        graph = b"".join(_ld(5, x) for x in inits) + b"".join(_ld(1, x) for x in nodes) + _ld(2, b"main_graph")
    else:
        graph = b"".join(_ld(1, x) for x in nodes) + _ld(2, b"main_graph") + b"".join(_ld(5, x) for x in inits)
    graph += _ld(11, value_info("input", ["batch_size", 3, h, w])) + _ld(12, value_info("output", ["batch_size", d]))
    return _vi(1, 6) + _ld(2, b"pytorch") + _ld(7, graph) + _ld(8, _ld(1, b"") + _vi(2, 11))
