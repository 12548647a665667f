"""SURVEY.md §8 f1, importer risk: the real `text-detection.onnx` / `text-recognition.onnx` (README.md:96-102) cannot be
loaded here, and until round 3 the importer had only ever read graphs written by this repo's own exporter.  These
tests hand-build the SAME networks in the other shapes PyTorch's ONNX exporter produces for the same model code
(operator set of ocrs/src/wasm_api.rs:35-56) and require each variant to import to exactly the `.ocrsm` bytes of the
canonical export — or to be refused with the offending node named:

  * U-Net skip padding with DYNAMIC amounts: Shape -> Gather -> Sub -> Div -> Unsqueeze -> Concat -> (ConstantOfShape,
    Concat, Reshape, Slice, Transpose, Reshape, Cast) -> Pad, as traced `F.pad(x1, [dx // 2, dx - dx // 2, ...])` with
    dynamic axes comes out, instead of constant-folded amounts;
  * opset 11 forms: `axes` of Unsqueeze / Squeeze as attributes (inputs since opset 13);
  * opset < 11 Pad (`pads` / `value` attributes) and opset < 10 Slice (`starts` / `ends` / `axes` attributes);
  * Linear as Add(bias, MatMul) — bias first;
  * GRU with both outputs named (`Y`, `Y_h`), `initial_h` sliced per layer out of one zeros(num_layers * 2, N, H)
    (torch's multi-layer lowering), layer outputs joined by Transpose + Reshape;
  * [N,C,1,W] -> [W,N,C] as permute(3,0,1,2) + reshape(W,N,-1) and as squeeze(2) + permute(2,0,1);
  * weights as Constant nodes instead of initializers, Identity nodes sprinkled in.
CPU only."""
import numpy as np
import pytest

from ocrs_amd import modelfile as mf
from ocrs_amd import onnx_pb as pb
from ocrs_amd.onnx_export import export_onnx
from ocrs_amd.onnx_import import OnnxImportError, import_onnx
from test_onnx_import import eval_onnx, small_detection, small_recognition

IN_DET = [("input", ["batch", 1, 100, 76])]
OUT_DET = [("output", ["batch", 1, "height", "width"])]
IN_REC = [("input", ["batch", 1, 32, "width"])]
OUT_REC = [("output", ["seq", "batch", "classes"])]


class Rewriter:
    """The canonical export, decoded, with helpers to splice other node sequences in."""

    def __init__(self, data):
        self.m = pb.parse_model(data)
        self.nodes = list(self.m.nodes)
        self.inits = dict(self.m.initializers)
        self.k = 0

    def name(self, hint):
        self.k += 1
        return "v_%s_%d" % (hint, self.k)

    def const(self, arr):
        out = self.name("const")
        return pb.Node("Constant", [], [out], {"value": np.asarray(arr)}, name=out), out

    def build(self, kind, opset=17):
        io = (IN_DET, OUT_DET) if kind == "det" else (IN_REC, OUT_REC)
        return pb.make_model(self.nodes, self.inits, io[0], io[1], opset=opset)

    def producer(self, value):
        for i, n in enumerate(self.nodes):
            if value in n.outputs:
                return i, n
        return None, None


def canon(kind):
    g = small_detection() if kind == "det" else small_recognition()
    return g, export_onnx(g)


def same_container(data, g):
    got = import_onnx(data)
    assert got.to_bytes() == g.to_bytes()


# ------------------------------------------------------------------ detection variants
def dynamic_pad_variant(data):
    """Every constant-folded pad vector [dx//2, dx - dx//2, dy//2, dy - dy//2] becomes shape arithmetic on the two
    tensors (what tracing with dynamic axes emits)."""
    r = Rewriter(data)
    n_rewritten = 0
    nodes = []
    for n in r.nodes:
        if n.op != "Pad":
            nodes.append(n)
            continue
        # walk back: Pad <- Cast <- Reshape <- Transpose <- Slice <- Reshape <- Concat(pads, ext); pads = Constant
        chain = []
        v = n.inputs[1]
        for _ in range(6):
            _, p = r.producer(v)
            chain.append(p)
            v = p.inputs[0]
        concat = chain[-1]
        assert concat.op == "Concat"
        _, pads_const = r.producer(concat.inputs[0])
        assert pads_const.op == "Constant"
        # the Concat that consumes this Pad tells us the skip tensor
        skip = next(c.inputs[0] for c in r.nodes if c.op == "Concat" and n.outputs[0] in c.inputs and c.attrs.get("axis") == 1)
        up = n.inputs[0]
        new = []

        def emit(op, inputs, attrs=None):
            o = r.name(op.lower())
            new.append(pb.Node(op, inputs, [o], attrs, name=o))
            return o

        def dim(t, ax):
            c, cn = r.const(np.array(ax, np.int64))
            new.append(c)
            return emit("Gather", [emit("Shape", [t]), cn], {"axis": 0})

        def c64(v):
            c, cn = r.const(np.array(v, np.int64))
            new.append(c)
            return cn

        dy = emit("Sub", [dim(skip, 2), dim(up, 2)])
        dx = emit("Sub", [dim(skip, 3), dim(up, 3)])
        two = c64(2)
        hy, hx = emit("Div", [dy, two]), emit("Div", [dx, two])
        ry, rx = emit("Sub", [dy, hy]), emit("Sub", [dx, hx])
        ax0 = c64([0])
        parts = [emit("Unsqueeze", [v, ax0]) for v in (hx, rx, hy, ry)]
        dyn = emit("Concat", parts, {"axis": 0})
        # splice: the old Constant's consumers read `dyn` instead
        idx = nodes.index(pads_const)
        nodes[idx:idx + 1] = new
        concat.inputs[0] = dyn
        nodes.append(n)
        n_rewritten += 1
    assert n_rewritten >= 2
    r.nodes = nodes
    return r.build("det")


def test_detection_dynamic_pad_amounts_import_to_the_same_container():
    g, data = canon("det")
    v = dynamic_pad_variant(data)
    ops = [n.op for n in pb.parse_model(v).nodes]
    assert ops.count("Sub") >= 8 and ops.count("Div") >= 4 and ops.count("Shape") >= 8
    same_container(v, g)


def test_detection_old_opset_pad_and_slice_attribute_forms():
    """opset 9/10-era lowering: Pad carries `pads` / `value` as attributes and the whole shape-arithmetic chain is
    folded away (the reversed Slice of torch's F.pad lowering only exists from opset 10 on, as inputs)."""
    g, data = canon("det")
    r = Rewriter(data)
    nodes = []
    for n in r.nodes:
        if n.op == "Pad":
            # evaluate the pads the canonical chain computes
            pv = eval_chain(r, n.inputs[1])
            nodes.append(pb.Node("Pad", [n.inputs[0]], n.outputs, {"mode": "constant", "pads": [int(x) for x in pv], "value": 0.0}, name=n.name))
        else:
            nodes.append(n)
    r.nodes = prune(nodes, "output")
    assert not any(n.op in ("ConstantOfShape", "Slice", "Cast") for n in r.nodes)
    same_container(r.build("det", opset=9), g)


def eval_chain(r, value):
    """numpy evaluation of a constant integer sub-graph of the canonical export (Constant / ConstantOfShape / Concat /
    Reshape / Slice(reverse) / Transpose / Cast)."""
    _, n = r.producer(value)
    if n is None:
        return np.asarray(r.inits[value])
    a = n.attrs
    ins = [eval_chain(r, i) for i in n.inputs if i]
    if n.op == "Constant":
        return np.asarray(a["value"])
    if n.op == "ConstantOfShape":
        return np.full(tuple(int(d) for d in ins[0]), np.asarray(a["value"]).reshape(-1)[0])
    if n.op == "Concat":
        return np.concatenate([np.atleast_1d(v) for v in ins], axis=a["axis"])
    if n.op == "Reshape":
        return np.reshape(ins[0], [int(v) for v in ins[1]])
    if n.op == "Slice":
        return ins[0][::-1]
    if n.op == "Transpose":
        return np.transpose(ins[0], a["perm"])
    if n.op == "Cast":
        return ins[0].astype(np.int64)
    raise AssertionError(n.op)


def prune(nodes, out_name):
    """drop nodes whose outputs nobody reads any more"""
    live, keep = {out_name}, []
    for n in reversed(nodes):
        if any(o in live for o in n.outputs):
            keep.append(n)
            live.update(i for i in n.inputs if i)
    return list(reversed(keep))


def test_detection_weights_as_constant_nodes_and_identities():
    g, data = canon("det")
    r = Rewriter(data)
    nodes = []
    for k, v in list(r.inits.items()):
        nodes.append(pb.Node("Constant", [], [k], {"value": np.asarray(v)}, name="const_" + k))
    r.inits = {}
    for n in r.nodes:
        nodes.append(n)
        if n.op == "MaxPool":   # an Identity between a pool and its consumers
            o = n.outputs[0]
            n.outputs[0] = o + "_pre"
            nodes.append(pb.Node("Identity", [o + "_pre"], [o], name="id_" + o))
    r.nodes = nodes
    same_container(r.build("det"), g)


# ------------------------------------------------------------------ recognition variants
def test_recognition_bias_first_named_yh_and_opset11_axes():
    g, data = canon("rec")
    r = Rewriter(data)
    n_add = n_gru = n_axes = 0
    for i, n in enumerate(r.nodes):
        if n.op == "Add" and any(k in r.inits for k in n.inputs):
            n.inputs = [n.inputs[1], n.inputs[0]]      # Add(bias, MatMul(x, W))
            n_add += 1
        if n.op == "GRU":
            n.outputs = [n.outputs[0], n.outputs[0] + "_h"]   # Y_h exists and is unused
            n_gru += 1
        if n.op in ("Unsqueeze", "Squeeze") and len(n.inputs) == 2:
            axes = eval_chain(r, n.inputs[1])
            r.nodes[i] = pb.Node(n.op, [n.inputs[0]], n.outputs, {"axes": [int(a) for a in np.atleast_1d(axes)]}, name=n.name)
            n_axes += 1
    assert n_add == 1 and n_gru == 2 and n_axes >= 3
    r.nodes = prune(r.nodes, "output")
    v = r.build("rec", opset=11)
    same_container(v, g)
    # and it still means the same thing (independent evaluator) — bias order is not observable
    x = np.random.default_rng(0).normal(0, 0.3, (2, 1, 32, 64)).astype(np.float32)
    r2 = Rewriter(data)
    for n in r2.nodes:
        if n.op == "Add" and any(k in r2.inits for k in n.inputs):
            n.inputs = [n.inputs[1], n.inputs[0]]
    a, b = eval_onnx(data, x), eval_onnx(r2.build("rec"), x)
    assert np.array_equal(a, b)


def test_recognition_initial_h_sliced_from_one_zeros_tensor_per_layer():
    """torch lowers a 2-layer bidirectional GRU with h0 = zeros(4, N, H) to per-layer Slice(h0, 2l, 2l + 2)."""
    g, data = canon("rec")
    r = Rewriter(data)
    grus = [n for n in r.nodes if n.op == "GRU"]
    assert len(grus) == 2
    H = grus[0].attrs["hidden_size"]
    nodes, first = [], True
    h0_all = None
    for n in r.nodes:
        if n.op == "GRU":
            layer = grus.index(n)
            new = []

            def emit(op, inputs, attrs=None):
                o = r.name(op.lower())
                new.append(pb.Node(op, inputs, [o], attrs, name=o))
                return o

            def c64(v):
                c, cn = r.const(np.array(v, np.int64))
                new.append(c)
                return cn

            if h0_all is None:
                nb = emit("Unsqueeze", [emit("Gather", [emit("Shape", [n.inputs[0]]), c64(1)], {"axis": 0}), c64([0])])
                shape = emit("Concat", [c64([4]), nb, c64([H])], {"axis": 0})
                h0_all = emit("ConstantOfShape", [shape], {"value": np.zeros(1, np.float32)})
            h0 = emit("Slice", [h0_all, c64([2 * layer]), c64([2 * layer + 2]), c64([0])])
            n.inputs[5] = h0
            nodes.extend(new)
        nodes.append(n)
    r.nodes = prune(nodes, "output")
    same_container(r.build("rec"), g)


@pytest.mark.parametrize("form", ["permute3012_reshape", "squeeze_permute"])
def test_recognition_other_lowerings_of_the_sequence_reshape(form):
    g, _ = canon("rec")
    if form == "squeeze_permute":
        same_container(export_onnx(g, toseq="squeeze"), g)
        return
    data = export_onnx(g)
    r = Rewriter(data)
    # find the canonical  Reshape(x, [N,-1,W]) -> Transpose(perm 2,0,1)  and replace it by
    # Transpose(x, perm 3,0,1,2) -> Reshape([W, N, -1]) with the torch shape arithmetic
    ti = next(i for i, n in enumerate(r.nodes) if n.op == "Transpose" and list(n.attrs.get("perm", [])) == [2, 0, 1])
    tr = r.nodes[ti]
    _, rs = r.producer(tr.inputs[0])
    x = rs.inputs[0]
    new = []

    def emit(op, inputs, attrs=None, out=None):
        o = out or r.name(op.lower())
        new.append(pb.Node(op, inputs, [o], attrs, name=o))
        return o

    def c64(v):
        c, cn = r.const(np.array(v, np.int64))
        new.append(c)
        return cn

    p = emit("Transpose", [x], {"perm": [3, 0, 1, 2]})
    w = emit("Unsqueeze", [emit("Gather", [emit("Shape", [x]), c64(3)], {"axis": 0}), c64([0])])
    nb = emit("Unsqueeze", [emit("Gather", [emit("Shape", [x]), c64(0)], {"axis": 0}), c64([0])])
    tgt = emit("Concat", [w, nb, c64([-1])], {"axis": 0})
    emit("Reshape", [p, tgt], out=tr.outputs[0])
    r.nodes[ti:ti + 1] = new
    r.nodes = prune(r.nodes, "output")
    same_container(r.build("rec"), g)


# ------------------------------------------------------------------ refusals name the node
def test_unsupported_torch_exports_are_refused_with_the_node_named():
    g, data = canon("det")
    r = Rewriter(data)
    for n in r.nodes:
        if n.op == "Pad":
            n.attrs["mode"] = "reflect"
            bad = n.name
            break
    with pytest.raises(OnnxImportError, match="Pad node .*%s.*constant zero" % bad):
        import_onnx(r.build("det"))
    r = Rewriter(data)   # F.interpolate instead of ConvTranspose
    i = next(i for i, n in enumerate(r.nodes) if n.op == "ConvTranspose")
    ct = r.nodes[i]
    r.nodes[i] = pb.Node("Resize", [ct.inputs[0]], ct.outputs, {"mode": "nearest"}, name="up_resize")
    with pytest.raises(OnnxImportError, match="Resize node 'up_resize'.*not supported"):
        import_onnx(r.build("det"))
    g, data = canon("rec")
    r = Rewriter(data)
    gru = next(n for n in r.nodes if n.op == "GRU")
    gru.attrs["linear_before_reset"] = 0
    with pytest.raises(OnnxImportError, match="GRU node .*linear_before_reset"):
        import_onnx(r.build("rec"))
    r = Rewriter(data)   # a unidirectional GRU
    gru = next(n for n in r.nodes if n.op == "GRU")
    gru.attrs["direction"] = "forward"
    with pytest.raises(OnnxImportError, match="GRU node .*bidirectional"):
        import_onnx(r.build("rec"))
    r = Rewriter(data)   # LogSoftmax over the batch axis
    ls = next(n for n in r.nodes if n.op == "LogSoftmax")
    ls.attrs["axis"] = 1
    with pytest.raises(OnnxImportError, match="LogSoftmax node .*class axis"):
        import_onnx(r.build("rec"))
