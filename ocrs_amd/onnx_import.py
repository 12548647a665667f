"""ONNX -> fixed-graph model importer.

The reference ships its networks as `text-detection.rten` / `text-recognition.rten`,
converted from `text-detection.onnx` / `text-recognition.onnx` (README.md:96-102); the
operators those graphs use are listed in ocrs/src/wasm_api.rs:35-56:

    Add AveragePool Cast Concat ConstantOfShape Conv ConvTranspose GRU Gather LogSoftmax
    MatMul MaxPool Pad Relu Reshape Shape Sigmoid Slice Transpose Unsqueeze

This module reads such a file (no `onnx` package needed: `onnx_pb.py`) and lowers it to
the op list of `modelfile.Graph`, which is what the HIP executor runs.  It works by
evaluating the graph once at a sample input size:

  * shape arithmetic (Shape/Gather/Unsqueeze/Concat/Cast/Slice/ConstantOfShape/Reshape/
    Transpose/Add/... on integer tensors, `Constant` nodes) is folded with numpy;
  * activations carry a *view* — the list of logical axes (N,C,H,W / T,N,C / T,D,N,H) the
    ONNX value currently has — so Reshape/Transpose/Squeeze chains collapse into the
    executor's NHWC / [T,N,C] conventions (TOSEQ is emitted where the view becomes [W,N,C]);
  * Conv(+Relu), depthwise Conv, MaxPool, AveragePool, ConvTranspose 2x2/s2, Pad+Concat,
    Sigmoid, bidirectional GRU (ONNX gate order z,r,h -> r,z,n; linear_before_reset=1 is
    required: that is what PyTorch exports and what the executor computes), MatMul+Add and
    LogSoftmax map one-to-one to executor ops with the weights transposed into the
    accumulation-order layouts of DESIGN.md §4.

Anything else raises `OnnxImportError` naming the node.  The real files are not obtainable
offline, so the importer is validated on graphs written by `onnx_export.py` in the torch
exporter's idiom (tests/test_onnx_import.py) — parity with the real model files is UNPINNED
until those files can be loaded.
"""
import numpy as np

from . import modelfile as mf
from .onnx_pb import parse_model


class OnnxImportError(ValueError):
    pass


class Act:
    """An activation: executor slot + the logical axes of the ONNX value that lives in it."""
    __slots__ = ("slot", "dims", "pad")

    def __init__(self, slot, dims, pad=None):
        self.slot, self.dims, self.pad = slot, list(dims), pad

    @property
    def roles(self):
        return [r for r, _ in self.dims]

    @property
    def shape(self):
        return [s for _, s in self.dims]

    def size(self, role):
        return dict(self.dims)[role]


NCHW = ["N", "C", "H", "W"]


def _is_seq(a):
    r = a.roles
    return len(r) == 3 and r[0] in ("T", "W") and r[1] == "N" and r[2] == "C"


class _Importer:
    def __init__(self, model, sample):
        self.m = model
        self.ops = []
        self.n_slots = 1
        self.env = {}
        self.producer_op = {}   # slot -> index in self.ops of the op that wrote it
        self.consumers = {}
        for n in model.nodes:
            for i in n.inputs:
                if i:
                    self.consumers[i] = self.consumers.get(i, 0) + 1
        for name, _ in model.outputs:
            self.consumers[name] = self.consumers.get(name, 0) + 1
        for k, v in model.initializers.items():
            self.env[k] = v
        if len(model.inputs) != 1:
            raise OnnxImportError("expected one graph input, found %d" % len(model.inputs))
        name, dims = model.inputs[0]
        self.input_dims = dims
        self.env[name] = Act(0, list(zip(NCHW, sample)))

    def new_slot(self):
        self.n_slots += 1
        return self.n_slots - 1

    def emit(self, op):
        self.ops.append(op)
        self.producer_op[op.out] = len(self.ops) - 1
        return op.out

    def err(self, node, msg):
        raise OnnxImportError("%s node %r: %s" % (node.op, node.name or node.outputs[0], msg))

    def get(self, node, i, optional=False):
        if i >= len(node.inputs) or node.inputs[i] == "":
            if optional:
                return None
            self.err(node, "missing input %d" % i)
        k = node.inputs[i]
        if k not in self.env:
            self.err(node, "input %r is not defined (graph not topologically sorted?)" % k)
        return self.env[k]

    def const(self, node, i, optional=False):
        v = self.get(node, i, optional)
        if v is None:
            return None
        if isinstance(v, Act):
            self.err(node, "input %d must be a constant" % i)
        return v

    def nchw(self, node, i=0):
        a = self.get(node, i)
        if not isinstance(a, Act) or a.roles != NCHW:
            self.err(node, "input %d must be an NCHW activation (has %s)" % (i, getattr(a, "roles", "constant")))
        if a.pad is not None:
            self.err(node, "a padded tensor may only feed a channel Concat")
        return a

    def seq(self, node, i=0):
        a = self.get(node, i)
        if not isinstance(a, Act) or not _is_seq(a):
            self.err(node, "input %d must be a [T,N,C] activation (has %s)" % (i, getattr(a, "roles", "constant")))
        if a.roles[0] == "W":  # still the NHWC slot of the conv stack: [N,1,W,C] -> [T,N,C]
            out = self.emit(mf.Op(mf.OP_TOSEQ, a.slot, self.new_slot()))
            a.slot, a.dims = out, [("T", a.dims[0][1]), a.dims[1], a.dims[2]]
        return a

    # ------------------------------------------------------------------ node handlers
    def run(self):
        for node in self.m.nodes:
            h = getattr(self, "op_" + node.op, None)
            if h is None:
                self.err(node, "operator not supported by the importer")
            r = h(node)
            if r is not None:
                self.env[node.outputs[0]] = r
        name = self.m.outputs[0][0]
        out = self.env.get(name)
        if not isinstance(out, Act):
            raise OnnxImportError("graph output %r is not an activation" % name)
        if _is_seq(out):
            if out.roles[0] == "W":
                out = Act(self.emit(mf.Op(mf.OP_TOSEQ, out.slot, self.new_slot())), out.dims)
            kind = mf.KIND_RECOGNITION
        elif out.roles == NCHW and out.pad is None:
            kind = mf.KIND_DETECTION
        else:
            raise OnnxImportError("graph output has unsupported layout %s" % out.roles)
        dims = self.input_dims or [-1, 1, -1, -1]
        in_shape = [-1, 1, dims[2] if len(dims) == 4 and dims[2] > 0 else -1, dims[3] if len(dims) == 4 and dims[3] > 0 else -1]
        return mf.Graph(kind, in_shape, self.ops, self.n_slots, out.slot)

    # --- constants / shape arithmetic
    def op_Constant(self, n):
        if "value" in n.attrs:
            return np.asarray(n.attrs["value"])
        for k in ("value_int", "value_float", "value_ints", "value_floats"):
            if k in n.attrs:
                return np.asarray(n.attrs[k])
        self.err(n, "no value attribute")

    def op_Identity(self, n):
        return self.get(n, 0)

    def op_Shape(self, n):
        a = self.get(n, 0)
        return np.asarray(a.shape if isinstance(a, Act) else np.shape(a), np.int64)

    def op_Gather(self, n):
        return np.take(self.const(n, 0), self.const(n, 1), axis=n.attrs.get("axis", 0))

    def op_Cast(self, n):
        to = {1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_}.get(n.attrs.get("to"))
        if to is None:
            self.err(n, "unsupported target type %r" % n.attrs.get("to"))
        return self.const(n, 0).astype(to)

    def op_ConstantOfShape(self, n):
        v = n.attrs.get("value")
        v = np.zeros(1, np.float32) if v is None else np.asarray(v).reshape(-1)
        return np.full(tuple(int(d) for d in self.const(n, 0)), v[0], dtype=v.dtype)

    def op_Slice(self, n):
        x = self.const(n, 0)
        if len(n.inputs) > 1:
            starts, ends = self.const(n, 1), self.const(n, 2)
            axes, steps = self.const(n, 3, True), self.const(n, 4, True)
        else:
            starts, ends, axes, steps = n.attrs["starts"], n.attrs["ends"], n.attrs.get("axes"), None
        axes = range(len(starts)) if axes is None else axes
        steps = [1] * len(starts) if steps is None else steps
        sl = [slice(None)] * x.ndim
        for s, e, ax, st in zip(starts, ends, axes, steps):
            s, e, st = int(s), int(e), int(st)
            if st < 0 and e < -x.shape[ax]:
                e = None  # "down to and including element 0"
            elif st > 0 and e > x.shape[ax]:
                e = None
            sl[int(ax)] = slice(s, e, st)
        return x[tuple(sl)]

    def _binary(self, n, f):
        a, b = self.get(n, 0), self.get(n, 1)
        if isinstance(a, Act) or isinstance(b, Act):
            return None
        return f(a, b)

    def op_Sub(self, n):
        r = self._binary(n, np.subtract)
        return r if r is not None else self.err(n, "only constant operands are supported")

    def op_Mul(self, n):
        r = self._binary(n, np.multiply)
        return r if r is not None else self.err(n, "only constant operands are supported")

    def op_Div(self, n):
        r = self._binary(n, lambda a, b: a // b if np.issubdtype(np.asarray(a).dtype, np.integer) else a / b)
        return r if r is not None else self.err(n, "only constant operands are supported")

    def op_Unsqueeze(self, n):
        x = self.get(n, 0)
        axes = self.const(n, 1) if len(n.inputs) > 1 else n.attrs["axes"]
        axes = [int(a) for a in np.asarray(axes).reshape(-1)]
        if isinstance(x, Act):
            dims = list(x.dims)
            for ax in sorted(a % (len(dims) + 1) for a in axes):
                dims.insert(ax, ("one", 1))
            return Act(x.slot, dims, x.pad)
        for ax in sorted(axes):
            x = np.expand_dims(x, ax)
        return x

    def op_Squeeze(self, n):
        x = self.get(n, 0)
        axes = self.const(n, 1, True) if len(n.inputs) > 1 else n.attrs.get("axes")
        if isinstance(x, Act):
            axes = [i for i, (_, s) in enumerate(x.dims) if s == 1] if axes is None else [int(a) % len(x.dims) for a in axes]
            if any(x.dims[a][1] != 1 for a in axes):
                self.err(n, "squeezing a non-unit axis")
            return Act(x.slot, [d for i, d in enumerate(x.dims) if i not in axes], x.pad)
        return np.squeeze(x, None if axes is None else tuple(int(a) for a in axes))

    def op_Transpose(self, n):
        x = self.get(n, 0)
        perm = n.attrs.get("perm")
        if isinstance(x, Act):
            perm = perm or list(reversed(range(len(x.dims))))
            return Act(x.slot, [x.dims[p] for p in perm], x.pad)
        return np.transpose(x, perm)

    def op_Reshape(self, n):
        x, tgt = self.get(n, 0), [int(v) for v in self.const(n, 1)]
        if not isinstance(x, Act):
            shp = [np.shape(x)[i] if v == 0 else v for i, v in enumerate(tgt)]
            return np.reshape(x, shp)
        src = x.dims
        tgt = [src[i][1] if v == 0 else v for i, v in enumerate(tgt)]
        total = int(np.prod([s for _, s in src]))
        if -1 in tgt:
            known = -int(np.prod(tgt))
            tgt[tgt.index(-1)] = total // known
        if int(np.prod(tgt)) != total:
            self.err(n, "cannot reshape %s to %s" % (x.shape, tgt))
        dims, i = [], 0
        for t in tgt:
            while i < len(src) and src[i][1] == 1 and t != 1:
                i += 1  # drop unit axes
            if t == 1 and not (i < len(src) and src[i][1] == 1):
                dims.append(("one", 1))
                continue
            if i >= len(src):
                self.err(n, "cannot reshape %s to %s" % (x.shape, tgt))
            if src[i][1] == t:
                dims.append(src[i])
                i += 1
                continue
            group, prod = [], 1
            while i < len(src) and prod < t:
                group.append(src[i])
                prod *= src[i][1]
                i += 1
            if prod != t:
                self.err(n, "reshape %s -> %s splits an axis" % (x.shape, tgt))
            roles = [r for r, s in group if s != 1 or r in ("C",)]
            real = [r for r, s in group if s != 1]
            if real in (["C"], ["D", "Hh"]) or (real == [] and "C" in roles):
                dims.append(("C", t))
            elif len(real) == 1:
                dims.append((real[0], t))
            else:
                self.err(n, "reshape %s -> %s merges axes %s, which the executor's layouts cannot express" %
                         (x.shape, tgt, [r for r, _ in group]))
        if any(s != 1 for _, s in src[i:]):
            self.err(n, "cannot reshape %s to %s" % (x.shape, tgt))
        return Act(x.slot, dims, x.pad)

    def op_Concat(self, n):
        vals = [self.get(n, i) for i in range(len(n.inputs))]
        axis = n.attrs.get("axis", 0)
        if not any(isinstance(v, Act) for v in vals):
            return np.concatenate([np.atleast_1d(v) for v in vals], axis=axis)
        if len(vals) != 2 or not all(isinstance(v, Act) for v in vals) or axis not in (1, -3):
            self.err(n, "only a channel Concat of two NCHW activations [skip, upsampled] is supported")
        skip, up = vals
        if skip.roles != NCHW or up.roles != NCHW or skip.pad is not None:
            self.err(n, "Concat inputs must be NCHW activations, the first one unpadded")
        sh, sw = skip.size("H"), skip.size("W")
        if (up.size("H"), up.size("W")) != (sh, sw):
            self.err(n, "Concat inputs differ in spatial size after padding: %s vs %s" % (skip.shape, up.shape))
        top, left, bottom, right = up.pad or (0, 0, 0, 0)
        if top != (top + bottom) // 2 or left != (left + right) // 2:
            self.err(n, "Pad before Concat is not the centred F.pad(diff//2, diff - diff//2) the executor's PADCAT applies")
        out = self.emit(mf.Op(mf.OP_PADCAT, skip.slot, self.new_slot(), in1=up.slot))
        return Act(out, [("N", skip.size("N")), ("C", skip.size("C") + up.size("C")), ("H", sh), ("W", sw)])

    def op_Pad(self, n):
        x = self.get(n, 0)
        if not isinstance(x, Act) or x.roles != NCHW or x.pad is not None:
            self.err(n, "only NCHW activations can be padded")
        pads = self.const(n, 1) if len(n.inputs) > 1 else n.attrs["pads"]
        pads = [int(p) for p in pads]
        mode = n.attrs.get("mode", b"constant")
        value = self.const(n, 2, True)
        if (mode.decode() if isinstance(mode, bytes) else mode) != "constant" or (value is not None and float(np.asarray(value).reshape(-1)[0]) != 0.0):
            self.err(n, "only constant zero padding is supported")
        if len(pads) != 8 or pads[0] or pads[1] or pads[4] or pads[5] or min(pads) < 0:
            self.err(n, "only non-negative spatial padding is supported (pads=%s)" % pads)
        top, left, bottom, right = pads[2], pads[3], pads[6], pads[7]
        dims = [x.dims[0], x.dims[1], ("H", x.size("H") + top + bottom), ("W", x.size("W") + left + right)]
        return Act(x.slot, dims, (top, left, bottom, right))

    # --- layers
    def op_Conv(self, n):
        x = self.nchw(n)
        w, b = self.const(n, 1), self.const(n, 2, True)
        cout, cin_g, kh, kw = w.shape
        group = n.attrs.get("group", 1)
        pads = n.attrs.get("pads", [0, 0, 0, 0])
        if n.attrs.get("auto_pad", b"NOTSET") not in (b"NOTSET", "NOTSET"):
            self.err(n, "auto_pad is not supported")
        if list(n.attrs.get("strides", [1, 1])) != [1, 1] or list(n.attrs.get("dilations", [1, 1])) != [1, 1]:
            self.err(n, "only stride 1 / dilation 1 convolutions are supported")
        if kh % 2 == 0 or kw % 2 == 0 or list(pads) != [kh // 2, kw // 2, kh // 2, kw // 2]:
            self.err(n, "only odd kernels with 'same' padding are supported (kernel %dx%d pads %s)" % (kh, kw, pads))
        b = np.zeros(cout, np.float32) if b is None else b
        c = x.size("C")
        if group == c and cout == c and cin_g == 1 and (kh, kw) == (3, 3):  # depthwise (incl. the 1 -> 1 input conv)
            op = mf.Op(mf.OP_DWCONV3, x.slot, self.new_slot(), kh=3, kw=3, cin=c, cout=c,
                       weights=(np.transpose(w[:, 0], (1, 2, 0)), b))
        elif group == 1:
            if cin_g != c:
                self.err(n, "weight has %d input channels, activation has %d" % (cin_g, c))
            op = mf.Op(mf.OP_CONV, x.slot, self.new_slot(), kh=kh, kw=kw, cin=c, cout=cout,
                       weights=(np.transpose(w, (2, 3, 1, 0)), b))
        else:
            self.err(n, "grouped convolution other than depthwise 3x3 (group=%d, C=%d)" % (group, c))
        return Act(self.emit(op), [x.dims[0], ("C", cout), x.dims[2], x.dims[3]])

    def op_Relu(self, n):
        x = self.get(n, 0)
        if not isinstance(x, Act) or x.slot not in self.producer_op:
            self.err(n, "Relu must follow a convolution")
        op = self.ops[self.producer_op[x.slot]]
        if op.type not in (mf.OP_CONV, mf.OP_DWCONV3) or op.relu or self.consumers.get(n.inputs[0], 0) != 1:
            self.err(n, "Relu must be the only consumer of a Conv output (the executor fuses it)")
        op.relu = 1
        return x

    def _pool(self, n, t):
        x = self.nchw(n)
        k = list(n.attrs["kernel_shape"])
        if list(n.attrs.get("strides", k)) != k or any(n.attrs.get("pads", [0, 0, 0, 0])) or n.attrs.get("ceil_mode", 0):
            self.err(n, "only non-overlapping, unpadded, floor-mode pooling is supported")
        out = self.emit(mf.Op(t, x.slot, self.new_slot(), kh=k[0], kw=k[1]))
        return Act(out, [x.dims[0], x.dims[1], ("H", x.size("H") // k[0]), ("W", x.size("W") // k[1])])

    def op_MaxPool(self, n):
        return self._pool(n, mf.OP_MAXPOOL)

    def op_AveragePool(self, n):
        return self._pool(n, mf.OP_AVGPOOL)

    def op_ConvTranspose(self, n):
        x = self.nchw(n)
        w, b = self.const(n, 1), self.const(n, 2, True)
        cin, cout, kh, kw = w.shape
        if (kh, kw) != (2, 2) or list(n.attrs.get("strides", [1, 1])) != [2, 2] or any(n.attrs.get("pads", [0] * 4)) \
                or n.attrs.get("group", 1) != 1 or any(n.attrs.get("output_padding", [0, 0])):
            self.err(n, "only 2x2 stride-2 transposed convolutions are supported")
        if cin != x.size("C"):
            self.err(n, "weight has %d input channels, activation has %d" % (cin, x.size("C")))
        b = np.zeros(cout, np.float32) if b is None else b
        out = self.emit(mf.Op(mf.OP_CONVT2, x.slot, self.new_slot(), kh=2, kw=2, cin=cin, cout=cout,
                              weights=(np.transpose(w, (2, 3, 0, 1)), b)))
        return Act(out, [x.dims[0], ("C", cout), ("H", 2 * x.size("H")), ("W", 2 * x.size("W"))])

    def op_Sigmoid(self, n):
        x = self.nchw(n)
        return Act(self.emit(mf.Op(mf.OP_SIGMOID, x.slot, self.new_slot())), x.dims)

    def op_GRU(self, n):
        x = self.seq(n)
        W, R, B = self.const(n, 1), self.const(n, 2), self.const(n, 3, True)
        if self.get(n, 4, True) is not None:
            self.err(n, "sequence_lens is not supported")
        h0 = self.get(n, 5, True)
        if h0 is not None and (isinstance(h0, Act) or np.any(np.asarray(h0) != 0)):
            self.err(n, "a non-zero initial_h is not supported")
        H = n.attrs["hidden_size"]
        direction = n.attrs.get("direction", b"forward")
        direction = direction.decode() if isinstance(direction, bytes) else direction
        if direction != "bidirectional" or W.shape[0] != 2:
            self.err(n, "only bidirectional GRUs are supported")
        if n.attrs.get("linear_before_reset", 0) != 1:
            self.err(n, "linear_before_reset=0 (the ONNX default) is not what PyTorch exports nor what the executor computes")
        if any(k in n.attrs for k in ("activations", "clip")):
            self.err(n, "custom activations / clip are not supported")
        I = x.size("C")
        if W.shape != (2, 3 * H, I) or R.shape != (2, 3 * H, H):
            self.err(n, "weight shapes %s / %s do not match input %d, hidden %d" % (W.shape, R.shape, I, H))
        B = np.zeros((2, 6 * H), np.float32) if B is None else B

        def rzn(m):  # rows z,r,h -> columns r,z,n of the transposed matrix
            z, r, h = m[:H], m[H:2 * H], m[2 * H:]
            return np.concatenate([r, z, h], axis=0)

        ws = []
        for d in range(2):
            ws += [rzn(W[d]).T, rzn(B[d][:3 * H]), rzn(R[d]).T, rzn(B[d][3 * H:])]
        out = self.emit(mf.Op(mf.OP_GRU, x.slot, self.new_slot(), cin=I, cout=2 * H, hidden=H, weights=ws))
        self.env[n.outputs[0]] = Act(out, [("T", x.shape[0]), ("D", 2), ("N", x.size("N")), ("Hh", H)])
        return None

    def op_MatMul(self, n):
        x, w = self.seq(n), self.const(n, 1)
        if w.ndim != 2 or w.shape[0] != x.size("C"):
            self.err(n, "weight %s does not match %d input features" % (w.shape, x.size("C")))
        out = self.emit(mf.Op(mf.OP_LINEAR, x.slot, self.new_slot(), cin=w.shape[0], cout=w.shape[1],
                              weights=(w, np.zeros(w.shape[1], np.float32))))
        return Act(out, [x.dims[0], x.dims[1], ("C", w.shape[1])])

    def op_Add(self, n):
        r = self._binary(n, np.add)
        if r is not None:
            return r
        a, b = self.get(n, 0), self.get(n, 1)
        act, c, ai = (a, b, 0) if isinstance(a, Act) else (b, a, 1)
        if isinstance(c, Act) or act.slot not in self.producer_op:
            self.err(n, "only a bias Add after MatMul is supported")
        op = self.ops[self.producer_op[act.slot]]
        c = np.asarray(c, np.float32).reshape(-1)
        if op.type != mf.OP_LINEAR or np.any(op.weights[1] != 0) or c.size != op.cout or self.consumers.get(n.inputs[ai], 0) != 1:
            self.err(n, "only a bias Add directly after MatMul is supported")
        op.weights[1] = c.copy()
        return act

    def op_LogSoftmax(self, n):
        x = self.seq(n)
        if n.attrs.get("axis", -1) not in (-1, 2):
            self.err(n, "LogSoftmax must run over the class axis")
        return Act(self.emit(mf.Op(mf.OP_LOGSOFTMAX, x.slot, self.new_slot())), x.dims)


def _signature(g):
    return [(o.type, o.in0, o.in1, o.out, o.relu, o.kh, o.kw, o.cin, o.cout, o.hidden) for o in g.ops]


def import_onnx(data, input_hw=None):
    """ONNX bytes (or a path) -> `modelfile.Graph`.

    `input_hw` is needed only for graphs whose spatial input dims are symbolic AND whose
    structure depends on them (a U-Net's skip padding).  A graph with a symbolic width
    (the recognition model) is imported at two widths and must lower to the same op list."""
    if isinstance(data, str):
        with open(data, "rb") as f:
            data = f.read()
    model = parse_model(data)
    if not model.inputs:
        raise OnnxImportError("graph has no input")
    dims = model.inputs[0][1] or [-1, 1, -1, -1]
    if len(dims) != 4 or dims[1] not in (1, -1):
        raise OnnxImportError("expected a [N,1,H,W] input, found %s" % dims)
    fixed_h = dims[2] if dims[2] > 0 else (input_hw[0] if input_hw else None)
    fixed_w = dims[3] if dims[3] > 0 else (input_hw[1] if input_hw else None)
    samples = [[2, 1, fixed_h or 64, fixed_w or 208]]
    if fixed_w is None:
        samples.append([3, 1, fixed_h or 64, 304])
    graphs = [_Importer(model, s).run() for s in samples]
    if len(graphs) == 2 and _signature(graphs[0]) != _signature(graphs[1]):
        raise OnnxImportError("the graph lowers differently at different input widths; pass input_hw")
    g = graphs[0]
    if input_hw and g.kind == mf.KIND_DETECTION:
        g.input_shape[2], g.input_shape[3] = fixed_h or -1, fixed_w or -1
    return g


if __name__ == "__main__":
    import sys

    if len(sys.argv) != 3:
        sys.exit("usage: python -m ocrs_amd.onnx_import model.onnx model.ocrsm")
    g = import_onnx(sys.argv[1])
    g.save(sys.argv[2])
    print("%s: %s, %d ops, input %s" % (sys.argv[2], "recognition" if g.kind else "detection", len(g.ops), g.input_shape))
