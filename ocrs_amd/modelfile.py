"""Fixed-graph model files (`*.ocrsm`) for the HIP executor, and the seeded
synthetic-weight generators used while the real ocrs weights are unobtainable.

The reference loads `text-detection.rten` / `text-recognition.rten` through
`rten::Model::load_file` (ocrs-cli/src/models.rs:100-107); neither the files
(downloaded from S3, ocrs-cli/src/main.rs:305-309) nor the `.rten` schema are
available offline, so this module defines a small self-describing container
for the same information: a linear list of ops over numbered activation slots
plus one fp32 weight blob.  The op vocabulary is the subset of
ocrs/src/wasm_api.rs:35-56 that the two networks need once BatchNorm is folded
and the shape-arithmetic ops (Shape/Gather/Slice/Cast/ConstantOfShape/Pad/
Unsqueeze/Concat) are collapsed into PADCAT / TOSEQ.

Layout (little endian):
  header   72 B : magic "OCRSMDL1", u32 version, u32 kind, i64 input_shape[4]
                  (NCHW, -1 = symbolic), u32 n_ops, u32 n_slots, u32 out_slot,
                  u32 reserved, u64 blob_floats
  op table n_ops x 176 B : u32 type, i32 in0, in1, out, relu, kh, kw, cin, cout,
                  hidden, u32 n_w, u32 reserved, 8 x {u64 offset, u64 count}
  blob     blob_floats x f32

Weight layouts (the accumulation order they imply is the numeric spec of
DESIGN.md §4): CONV [KH][KW][Cin][Cout] + bias[Cout]; DWCONV3 [3][3][C] +
bias[C]; CONVT2 [2][2][Cin][Cout] + bias; LINEAR [K][O] + bias;
GRU per direction: Wi [I][3H], bi [3H], Wh [H][3H], bh [3H] (gate order r,z,n).
"""
import struct

import numpy as np

MAGIC = b"OCRSMDL1"
KIND_DETECTION, KIND_RECOGNITION = 0, 1

OP_CONV, OP_DWCONV3, OP_MAXPOOL, OP_AVGPOOL, OP_CONVT2, OP_PADCAT, OP_SIGMOID, OP_TOSEQ, OP_GRU, OP_LINEAR, \
    OP_LOGSOFTMAX = range(11)
OP_NAMES = ["conv", "dwconv3", "maxpool", "avgpool", "convt2", "padcat", "sigmoid", "toseq", "gru", "linear",
            "logsoftmax"]

_HDR = struct.Struct("<8sII4qIIIIQ")
_OP = struct.Struct("<I9iII16Q")
assert _HDR.size == 72 and _OP.size == 176


class Op:
    __slots__ = ("type", "in0", "in1", "out", "relu", "kh", "kw", "cin", "cout", "hidden", "weights")

    def __init__(self, type, in0, out, in1=-1, relu=0, kh=0, kw=0, cin=0, cout=0, hidden=0, weights=()):
        self.type, self.in0, self.in1, self.out = type, in0, in1, out
        self.relu, self.kh, self.kw, self.cin, self.cout, self.hidden = relu, kh, kw, cin, cout, hidden
        self.weights = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]

    def __repr__(self):
        return "Op(%s %d,%d->%d k=%dx%d c=%d->%d h=%d relu=%d)" % (
            OP_NAMES[self.type], self.in0, self.in1, self.out, self.kh, self.kw, self.cin, self.cout, self.hidden,
            self.relu)


class Graph:
    def __init__(self, kind, input_shape, ops, n_slots, out_slot):
        self.kind = kind
        self.input_shape = list(input_shape)  # NCHW, -1 symbolic
        self.ops = ops
        self.n_slots = n_slots
        self.out_slot = out_slot

    # ---- serialisation
    def to_bytes(self):
        blob = []
        off = 0
        table = b""
        for op in self.ops:
            refs = []
            for w in op.weights:
                refs += [off, w.size]
                blob.append(w.ravel())
                off += w.size
            refs += [0] * (16 - len(refs))
            table += _OP.pack(op.type, op.in0, op.in1, op.out, op.relu, op.kh, op.kw, op.cin, op.cout, op.hidden,
                              len(op.weights), 0, *refs)
        hdr = _HDR.pack(MAGIC, 1, self.kind, *self.input_shape, len(self.ops), self.n_slots, self.out_slot, 0, off)
        data = np.concatenate(blob).astype("<f4").tobytes() if blob else b""
        return hdr + table + data

    def save(self, path):
        with open(path, "wb") as f:
            f.write(self.to_bytes())

    @staticmethod
    def from_bytes(buf):
        magic, version, kind, n, c, h, w, n_ops, n_slots, out_slot, _, blob_floats = _HDR.unpack_from(buf, 0)
        if magic != MAGIC or version != 1:
            raise ValueError("not an OCRSMDL1 model file")
        blob = np.frombuffer(buf, dtype="<f4", count=blob_floats, offset=_HDR.size + n_ops * _OP.size)
        ops = []
        for i in range(n_ops):
            f = _OP.unpack_from(buf, _HDR.size + i * _OP.size)
            t, in0, in1, out, relu, kh, kw, cin, cout, hidden, n_w = f[:11]
            refs = f[12:]
            ws = [blob[refs[2 * j]:refs[2 * j] + refs[2 * j + 1]] for j in range(n_w)]
            op = Op(t, in0, out, in1, relu, kh, kw, cin, cout, hidden)
            op.weights = _reshape_weights(op, ws)
            ops.append(op)
        return Graph(kind, [n, c, h, w], ops, n_slots, out_slot)

    @staticmethod
    def load(path):
        with open(path, "rb") as f:
            return Graph.from_bytes(f.read())

    # ---- bookkeeping used by bench.py / DESIGN.md
    def flops(self, n, h, w):
        """Algorithmic FLOPs (2*MAC) of one forward at batch n, input h x w,
        per SURVEY.md §8(d): conv 2*N*Ho*Wo*Cout*(Cin/groups)*Kh*Kw; convT same with
        input spatial; GRU per layer/dir 2*T*N*3H*(I+H); linear 2*rows*I*O."""
        shapes = {0: (n, h, w, 1)}
        total = 0
        per_op = []
        for op in self.ops:
            s = shapes[op.in0]
            f = 0
            if op.type == OP_CONV:
                o = s[:3] + (op.cout,)
                f = 2 * s[0] * s[1] * s[2] * op.cout * op.cin * op.kh * op.kw
            elif op.type == OP_DWCONV3:
                o = s
                f = 2 * s[0] * s[1] * s[2] * s[3] * 9
            elif op.type in (OP_MAXPOOL, OP_AVGPOOL):
                o = (s[0], s[1] // op.kh, s[2] // op.kw, s[3])
            elif op.type == OP_CONVT2:
                o = (s[0], 2 * s[1], 2 * s[2], op.cout)
                f = 2 * s[0] * s[1] * s[2] * op.cin * op.cout * 4
            elif op.type == OP_PADCAT:
                k = shapes[op.in0]
                o = k[:3] + (k[3] + shapes[op.in1][3],)
            elif op.type in (OP_SIGMOID, OP_LOGSOFTMAX):
                o = s
            elif op.type == OP_TOSEQ:
                o = (s[2], s[0], s[3])  # [T,N,C]
            elif op.type == OP_GRU:
                o = (s[0], s[1], 2 * op.hidden)
                f = 2 * 2 * s[0] * s[1] * 3 * op.hidden * (op.cin + op.hidden)
            elif op.type == OP_LINEAR:
                o = s[:-1] + (op.cout,)
                f = 2 * int(np.prod(s[:-1])) * op.cin * op.cout
            shapes[op.out] = o
            per_op.append((OP_NAMES[op.type], f, o))
            total += f
        return total, per_op, shapes[self.out_slot]


def _reshape_weights(op, ws):
    t = op.type
    if t == OP_CONV:
        return [ws[0].reshape(op.kh, op.kw, op.cin, op.cout), ws[1]]
    if t == OP_DWCONV3:
        return [ws[0].reshape(3, 3, op.cin), ws[1]]
    if t == OP_CONVT2:
        return [ws[0].reshape(2, 2, op.cin, op.cout), ws[1]]
    if t == OP_LINEAR:
        return [ws[0].reshape(op.cin, op.cout), ws[1]]
    if t == OP_GRU:
        h, i = op.hidden, op.cin
        out = []
        for d in range(2):
            wi, bi, wh, bh = ws[4 * d:4 * d + 4]
            out += [wi.reshape(i, 3 * h), bi, wh.reshape(h, 3 * h), bh]
        return out
    return list(ws)


# --------------------------------------------------------------------------
# Synthetic architectures.  [UNVERIFIED-RECALL of robertknight/ocrs-models —
# SURVEY.md §2.4.]  Real weights are not obtainable offline; everything
# measured or tested with these files is labelled "synthetic weights".
# --------------------------------------------------------------------------
class _Builder:
    def __init__(self, rng):
        self.rng = rng
        self.ops = []
        self.n_slots = 1

    def slot(self):
        self.n_slots += 1
        return self.n_slots - 1

    def he(self, shape, fan_in, gain=1.0):
        return (self.rng.standard_normal(shape) * (gain * np.sqrt(2.0 / fan_in))).astype(np.float32)

    def small(self, n, scale=0.05):
        return (self.rng.standard_normal(n) * scale).astype(np.float32)

    def conv(self, x, cin, cout, k=1, relu=0, w=None, b=None, gain=1.0):
        w = self.he((k, k, cin, cout), k * k * cin, gain) if w is None else w
        b = self.small(cout) if b is None else b
        o = self.slot()
        self.ops.append(Op(OP_CONV, x, o, relu=relu, kh=k, kw=k, cin=cin, cout=cout, weights=(w, b)))
        return o

    def dwconv(self, x, c, w=None, b=None):
        w = self.he((3, 3, c), 9, 0.7) if w is None else w
        b = self.small(c) if b is None else b
        o = self.slot()
        self.ops.append(Op(OP_DWCONV3, x, o, kh=3, kw=3, cin=c, cout=c, weights=(w, b)))
        return o

    def pool(self, x, kh, kw, avg=False):
        o = self.slot()
        self.ops.append(Op(OP_AVGPOOL if avg else OP_MAXPOOL, x, o, kh=kh, kw=kw))
        return o

    def simple(self, t, x, **kw):
        o = self.slot()
        self.ops.append(Op(t, x, o, **kw))
        return o


def build_detection(in_hw=(800, 600), depths=(8, 16, 32, 32, 64, 128, 256), seed=1, ink_level=0.0, ink_gain=1.0, ink_sign=1):
    """U-Net of depthwise-separable DoubleConvs (ocrs-models `DetectionModel`,
    recollected): in_conv; Down = MaxPool2 -> DoubleConv; Up = ConvTranspose2x2/s2
    -> pad to skip -> concat [skip, up] -> DoubleConv; 1x1 conv -> Sigmoid.

    The level-0 skip path is hand-set so that channel 0 carries a blurred
    "darkness" feature through to the output logit (dark blobs on a light page
    -> probability ~0.98, background ~0.02); every other weight is seeded
    random and contributes a small perturbation.  This gives synthetic pages a
    realistic component count for the post-processing stages.

    `ink_level` / `ink_gain` (defaults: the file every fixture was made with) move the hand-set feature's operating
    point: darkness = relu(-2 * ink_sign * (blur(x) - ink_level)), logit = 8 * ink_gain * darkness - 4 (ink_sign = -1:
    light text on a dark page).  With ink_level ~0.3 the
    pale, anti-aliased strokes of real text scaled down to the detection input (scanned pages, screenshots) are marked
    as well, so the masks follow the glyphs — the natural-image fixtures of tests/golden/reference use such a file."""
    rng = np.random.default_rng(seed)
    B = _Builder(rng)

    def double_conv(x, cin, cout, carry=False, first=False):
        # DepthwiseConv(cin->cout) + BN(folded) + ReLU, twice.
        w_dw1 = B.he((3, 3, cin), 9, 0.7)
        w_pw1 = B.he((1, 1, cin, cout), cin)
        b_pw1 = B.small(cout)
        w_dw2 = B.he((3, 3, cout), 9, 0.7)
        w_pw2 = B.he((1, 1, cout, cout), cout)
        b_pw2 = B.small(cout)
        b_dw1, b_dw2 = B.small(cin), B.small(cout)
        if carry:
            if first:  # input channel 0 is the grey page in [-0.5, 0.5]
                w_dw1[:, :, 0] = 1.0 / 9.0
                b_dw1[0] = 0.0
                w_pw1[0, 0, :, 0] = 0.0
                w_pw1[0, 0, 0, 0] = np.float32(-2.0 * ink_sign)  # darkness = relu(-2 * ink_sign * (blur(x) - ink_level))
                b_pw1[0] = np.float32(2.0 * ink_sign * ink_level)
                w_dw2[:, :, 0] = 1.0 / 9.0
            else:
                w_dw1[:, :, 0] = 0.0
                w_dw1[1, 1, 0] = 1.0
                b_dw1[0] = 0.0
                w_pw1[0, 0, :, 0] *= 0.02
                w_pw1[0, 0, 0, 0] = 1.0
                b_pw1[0] = 0.0
                w_dw2[:, :, 0] = 0.0
                w_dw2[1, 1, 0] = 1.0
            b_dw2[0] = 0.0
            w_pw2[0, 0, :, 0] *= 0.02
            w_pw2[0, 0, 0, 0] = 1.0
            b_pw2[0] = 0.0
        x = B.dwconv(x, cin, w_dw1, b_dw1)
        x = B.conv(x, cin, cout, 1, relu=1, w=w_pw1, b=b_pw1)
        x = B.dwconv(x, cout, w_dw2, b_dw2)
        x = B.conv(x, cout, cout, 1, relu=1, w=w_pw2, b=b_pw2)
        return x

    skips = [double_conv(0, 1, depths[0], carry=True, first=True)]
    for i in range(len(depths) - 1):
        p = B.pool(skips[-1], 2, 2)
        skips.append(double_conv(p, depths[i], depths[i + 1]))
    x = skips[-1]
    for i in reversed(range(len(depths) - 1)):
        cin, cout = depths[i + 1], depths[i]
        w = B.he((2, 2, cin, cout), cin)
        o = B.slot()
        B.ops.append(Op(OP_CONVT2, x, o, kh=2, kw=2, cin=cin, cout=cout, weights=(w, B.small(cout))))
        c = B.slot()
        B.ops.append(Op(OP_PADCAT, skips[i], c, in1=o))
        x = double_conv(c, 2 * cout, cout, carry=(i == 0))
    w_out = B.he((1, 1, depths[0], 1), depths[0], 0.15)
    w_out[0, 0, 0, 0] = np.float32(8.0 * ink_gain)
    x = B.conv(x, depths[0], 1, 1, relu=0, w=w_out, b=np.array([-4.0], np.float32))
    x = B.simple(OP_SIGMOID, x)
    return Graph(KIND_DETECTION, [-1, 1, in_hw[0], in_hw[1]], B.ops, B.n_slots, x)


def build_recognition(n_classes=97, in_h=64, seed=2, hidden=256, chans=(32, 64, 128, 128, 128, 128)):
    """CRNN (ocrs-models `RecognitionModel`, recollected): six 3x3 Conv+ReLU
    (BN folded) with MaxPool (2,2),(2,2),-,(2,1),-,(2,1); AvgPool (4,1);
    [N,1,W/4,C] -> [T,N,C]; 2-layer bidirectional GRU(hidden); Linear ->
    LogSoftmax.  Output [T=W/4, N, n_classes] (recognition.rs:399-401)."""
    rng = np.random.default_rng(seed)
    B = _Builder(rng)
    pools = [(2, 2), (2, 2), None, (2, 1), None, (2, 1)]
    x, cin, h = 0, 1, in_h
    for cout, pool in zip(chans, pools):
        x = B.conv(x, cin, cout, 3, relu=1, gain=1.3 if cin == 1 else 1.0)
        if pool:
            x = B.pool(x, *pool)
            h //= pool[0]
        cin = cout
    x = B.pool(x, h, 1, avg=True)
    x = B.simple(OP_TOSEQ, x)
    feat = cin
    for layer in range(2):
        ws = []
        for d in range(2):
            k = 1.0 / np.sqrt(hidden)
            ws += [rng.uniform(-k, k, (feat, 3 * hidden)).astype(np.float32) * (2.0 if layer == 0 else 1.0),
                   rng.uniform(-k, k, 3 * hidden).astype(np.float32),
                   rng.uniform(-k, k, (hidden, 3 * hidden)).astype(np.float32),
                   rng.uniform(-k, k, 3 * hidden).astype(np.float32)]
        o = B.slot()
        B.ops.append(Op(OP_GRU, x, o, cin=feat, cout=2 * hidden, hidden=hidden, weights=ws))
        x, feat = o, 2 * hidden
    w = (rng.standard_normal((feat, n_classes)) * (6.0 / np.sqrt(feat))).astype(np.float32)
    b = np.zeros(n_classes, np.float32)
    b[0] = 2.5  # favour CTC blank so decoded lines have a plausible length
    o = B.slot()
    B.ops.append(Op(OP_LINEAR, x, o, cin=feat, cout=n_classes, weights=(w, b)))
    x = B.simple(OP_LOGSOFTMAX, o)
    return Graph(KIND_RECOGNITION, [-1, 1, in_h, -1], B.ops, B.n_slots, x)


def calibrate_recognition_head(graph, run, crops_nchw, blank_boost=1.0):
    """Random CRNN weights decode to one or two classes whatever the input.  Shift
    the final Linear bias by the per-class mean log-probability measured on a
    sample batch so that the arg-max follows the input; `run(graph_bytes, nchw)`
    -> [T,N,C] log-probs may be the HIP executor or any other executor of the
    container (the result is just another synthetic weight file)."""
    lp = np.asarray(run(graph.to_bytes(), crops_nchw), dtype=np.float64)
    mean = lp.reshape(-1, lp.shape[-1]).mean(axis=0)
    lin = [op for op in graph.ops if op.type == OP_LINEAR][-1]
    b = lin.weights[1].astype(np.float64) - (mean - mean.mean())
    b[0] += blank_boost
    lin.weights[1] = b.astype(np.float32)
    return graph
