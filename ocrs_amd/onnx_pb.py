"""Minimal protobuf wire codec for the part of the ONNX schema the OCR models use.

There is no `onnx` package (and no network) in this environment, so the importer
(`onnx_import.py`) and the exporter (`onnx_export.py`) read and
write `ModelProto` directly.  Field numbers follow onnx/onnx.proto3:

  ModelProto      ir_version=1 producer_name=2 graph=7 opset_import=8
  OperatorSetId   domain=1 version=2
  GraphProto      node=1 name=2 initializer=5 input=11 output=12
  NodeProto       input=1 output=2 name=3 op_type=4 attribute=5
  AttributeProto  name=1 f=2 i=3 s=4 t=5 floats=7 ints=8 type=20
  TensorProto     dims=1 data_type=2 float_data=4 int32_data=5 int64_data=7 name=8 raw_data=9
  ValueInfoProto  name=1 type=2 ; TypeProto tensor_type=1 ; Tensor elem_type=1 shape=2
  TensorShapeProto dim=1 ; Dimension dim_value=1 dim_param=2
"""
import struct

import numpy as np

FLOAT, INT32, INT64, BOOL = 1, 6, 7, 9
_NP = {FLOAT: np.dtype("<f4"), INT32: np.dtype("<i4"), INT64: np.dtype("<i8"), BOOL: np.dtype("?")}
ATTR_FLOAT, ATTR_INT, ATTR_STRING, ATTR_TENSOR, ATTR_FLOATS, ATTR_INTS = 1, 2, 3, 4, 6, 7


# ----------------------------------------------------------------------------- wire level
def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def fields(buf):
    """Yield (field_number, wire_type, value) of one message; length-delimited values are memoryviews."""
    buf = memoryview(buf)
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


def _packed_varints(v, wt):
    if wt == 0:
        return [_signed(v)]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed(x))
    return out


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(num, wt):
    return _enc_varint(num << 3 | wt)


def enc_int(num, v):
    return _key(num, 0) + _enc_varint(v)


def enc_bytes(num, b):
    if isinstance(b, str):
        b = b.encode()
    return _key(num, 2) + _enc_varint(len(b)) + bytes(b)


def enc_float(num, f):
    return _key(num, 5) + struct.pack("<f", f)


# ----------------------------------------------------------------------------- decode
def parse_tensor(buf):
    dims, dtype, name, raw = [], FLOAT, "", None
    f32, i32, i64 = [], [], []
    for num, wt, v in fields(buf):
        if num == 1:
            dims += _packed_varints(v, wt)
        elif num == 2:
            dtype = v
        elif num == 4:
            f32 += list(np.frombuffer(v, "<f4")) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif num == 5:
            i32 += _packed_varints(v, wt)
        elif num == 7:
            i64 += _packed_varints(v, wt)
        elif num == 8:
            name = bytes(v).decode()
        elif num == 9:
            raw = bytes(v)
    if dtype not in _NP:
        raise ValueError("tensor %r: unsupported ONNX data type %d" % (name, dtype))
    if raw is not None:
        arr = np.frombuffer(raw, _NP[dtype])
    elif dtype == FLOAT:
        arr = np.asarray(f32, "<f4")
    elif dtype == INT64:
        arr = np.asarray(i64, "<i8")
    else:
        arr = np.asarray(i32).astype(_NP[dtype])
    return name, arr.reshape(dims).copy()


def parse_attribute(buf):
    name, val, ints, floats, typ = "", None, [], [], 0
    for num, wt, v in fields(buf):
        if num == 1:
            name = bytes(v).decode()
        elif num == 2:
            val = struct.unpack("<f", v)[0]
        elif num == 3:
            val = _signed(v)
        elif num == 4:
            val = bytes(v)
        elif num == 5:
            val = parse_tensor(v)[1]
        elif num == 7:
            floats += list(np.frombuffer(v, "<f4")) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif num == 8:
            ints += _packed_varints(v, wt)
        elif num == 20:
            typ = v
    if typ == ATTR_INTS or (val is None and ints):
        val = ints
    elif typ == ATTR_FLOATS or (val is None and floats):
        val = floats
    elif val is None and typ == ATTR_INTS:
        val = []
    return name, val


class Node:
    __slots__ = ("op", "name", "inputs", "outputs", "attrs")

    def __init__(self, op, inputs, outputs, attrs=None, name=""):
        self.op, self.inputs, self.outputs, self.attrs, self.name = op, list(inputs), list(outputs), dict(attrs or {}), name

    def __repr__(self):
        return "%s(%s -> %s)" % (self.op, ",".join(self.inputs), ",".join(self.outputs))


def parse_node(buf):
    n = Node("", [], [])
    for num, wt, v in fields(buf):
        if num == 1:
            n.inputs.append(bytes(v).decode())
        elif num == 2:
            n.outputs.append(bytes(v).decode())
        elif num == 3:
            n.name = bytes(v).decode()
        elif num == 4:
            n.op = bytes(v).decode()
        elif num == 5:
            k, a = parse_attribute(v)
            n.attrs[k] = a
    return n


def parse_value_info(buf):
    name, dims = "", None
    for num, wt, v in fields(buf):
        if num == 1:
            name = bytes(v).decode()
        elif num == 2:
            for n2, _, v2 in fields(v):
                if n2 == 1:  # tensor_type
                    for n3, _, v3 in fields(v2):
                        if n3 == 2:  # shape
                            dims = []
                            for n4, _, v4 in fields(v3):
                                if n4 == 1:
                                    d = -1
                                    for n5, _, v5 in fields(v4):
                                        if n5 == 1:
                                            d = _signed(v5)
                                    dims.append(d)
    return name, dims


class Model:
    """Decoded ModelProto: nodes in file order, initializers by name, graph inputs/outputs with dims (-1 symbolic)."""

    def __init__(self):
        self.nodes, self.initializers, self.inputs, self.outputs = [], {}, [], []
        self.opset, self.ir_version, self.producer = 0, 0, ""


def parse_model(data):
    m = Model()
    graph = None
    for num, wt, v in fields(data):
        if num == 1:
            m.ir_version = v
        elif num == 2:
            m.producer = bytes(v).decode()
        elif num == 7:
            graph = v
        elif num == 8:
            dom, ver = "", 0
            for n2, _, v2 in fields(v):
                if n2 == 1:
                    dom = bytes(v2).decode()
                elif n2 == 2:
                    ver = v2
            if dom in ("", "ai.onnx"):
                m.opset = ver
    if graph is None:
        raise ValueError("not an ONNX ModelProto: no graph")
    for num, wt, v in fields(graph):
        if num == 1:
            m.nodes.append(parse_node(v))
        elif num == 5:
            name, arr = parse_tensor(v)
            m.initializers[name] = arr
        elif num == 11:
            m.inputs.append(parse_value_info(v))
        elif num == 12:
            m.outputs.append(parse_value_info(v))
    m.inputs = [(n, d) for n, d in m.inputs if n not in m.initializers]
    return m


# ----------------------------------------------------------------------------- encode
def make_tensor(name, arr):
    arr = np.asarray(arr)
    if arr.dtype.kind == "f":
        dt, arr = FLOAT, arr.astype("<f4")
    elif arr.dtype.kind in "iu":
        dt, arr = INT64, arr.astype("<i8")
    elif arr.dtype.kind == "b":
        dt = BOOL
    else:
        raise ValueError("unsupported dtype %s" % arr.dtype)
    out = b"".join(enc_int(1, d) for d in arr.shape) + enc_int(2, dt)
    if name:
        out += enc_bytes(8, name)
    return out + enc_bytes(9, np.ascontiguousarray(arr).tobytes())


def make_attribute(name, val):
    out = enc_bytes(1, name)
    if isinstance(val, float):
        return out + enc_float(2, val) + enc_int(20, ATTR_FLOAT)
    if isinstance(val, (int, np.integer)):
        return out + enc_int(3, int(val)) + enc_int(20, ATTR_INT)
    if isinstance(val, (bytes, str)):
        return out + enc_bytes(4, val) + enc_int(20, ATTR_STRING)
    if isinstance(val, np.ndarray):
        return out + enc_bytes(5, make_tensor("", val)) + enc_int(20, ATTR_TENSOR)
    if isinstance(val, (list, tuple)):
        if val and isinstance(val[0], float):
            return out + b"".join(enc_float(7, f) for f in val) + enc_int(20, ATTR_FLOATS)
        return out + enc_bytes(8, b"".join(_enc_varint(int(i)) for i in val)) + enc_int(20, ATTR_INTS)
    raise ValueError("unsupported attribute %r" % (val,))


def make_node(node):
    out = b"".join(enc_bytes(1, i) for i in node.inputs) + b"".join(enc_bytes(2, o) for o in node.outputs)
    if node.name:
        out += enc_bytes(3, node.name)
    out += enc_bytes(4, node.op)
    return out + b"".join(enc_bytes(5, make_attribute(k, v)) for k, v in node.attrs.items())


def make_value_info(name, dims):
    shape = b""
    for d in dims:
        shape += enc_bytes(1, enc_bytes(2, d) if isinstance(d, str) else enc_int(1, d))
    ttype = enc_int(1, FLOAT) + enc_bytes(2, shape)
    return enc_bytes(1, name) + enc_bytes(2, enc_bytes(1, ttype))


def make_model(nodes, initializers, inputs, outputs, opset=17, name="ocrs_amd", producer="ocrs_amd"):
    g = b"".join(enc_bytes(1, make_node(n)) for n in nodes) + enc_bytes(2, name)
    g += b"".join(enc_bytes(5, make_tensor(k, v)) for k, v in initializers.items())
    g += b"".join(enc_bytes(11, make_value_info(n, d)) for n, d in inputs)
    g += b"".join(enc_bytes(12, make_value_info(n, d)) for n, d in outputs)
    return enc_int(1, 8) + enc_bytes(2, producer) + enc_bytes(7, g) + enc_bytes(8, enc_bytes(1, "") + enc_int(2, opset))
