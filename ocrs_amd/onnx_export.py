"""Write a fixed-graph model (`modelfile.Graph`) as an ONNX file in the shape PyTorch's
exporter gives the ocrs-models networks (BatchNorm folded; the operator set is exactly the
one ocrs/src/wasm_api.rs:35-56 registers for the real models).

Two uses:
  * the round-trip tests of the importer — no real `text-detection.onnx` /
    `text-recognition.onnx` is obtainable offline, so the importer is exercised on graphs that
    use the same operators and the same shape-arithmetic idioms (torch's `F.pad` lowering:
    ConstantOfShape/Concat/Reshape/Slice/Transpose/Cast -> Pad; `reshape`/`permute` lowering:
    Shape/Gather/Unsqueeze/Concat -> Reshape -> Transpose; zero `initial_h` via ConstantOfShape);
  * cross-checking against the reference itself on a machine that has Rust: export the synthetic
    weights, `rten-convert` them, run ocrs with `--detect-model/--rec-model`, compare
    (INTEGRATION.md §5).
"""
import numpy as np

from . import modelfile as mf
from .onnx_pb import Node, make_model

INT64_MIN = -9223372036854775807


class _Emitter:
    def __init__(self):
        self.nodes, self.inits, self.n = [], {}, 0

    def name(self, hint):
        self.n += 1
        return "%s_%d" % (hint, self.n)

    def const(self, arr, hint="c"):
        k = self.name(hint)
        self.inits[k] = np.asarray(arr)
        return k

    def node(self, op, inputs, attrs=None, hint=None):
        out = self.name(hint or op.lower())
        self.nodes.append(Node(op, inputs, [out], attrs, name=out))
        return out

    def constant_node(self, arr):
        """Constant as a node (what the torch exporter emits for Python scalars/lists)."""
        return self.node("Constant", [], {"value": np.asarray(arr)}, hint="constant")


def export_onnx(graph, in_hw=None, opset=17, toseq="reshape"):
    """Serialise `graph` to ONNX bytes.  `in_hw` fixes the spatial input size used for the static
    U-Net pad amounts (defaults to the graph's own fixed dims; recognition graphs need none).
    `toseq`: how [N,C,1,W] becomes [W,N,C] — "reshape" (x.reshape(N,-1,W).permute(2,0,1), with the
    Shape/Gather/Unsqueeze/Concat arithmetic torch emits) or "squeeze" (x.squeeze(2).permute(2,0,1))."""
    E = _Emitter()
    h0 = graph.input_shape[2] if graph.input_shape[2] > 0 else (in_hw or (0, 0))[0]
    w0 = graph.input_shape[3] if graph.input_shape[3] > 0 else (in_hw or (0, 0))[1]
    val = {0: "input"}
    hw = {0: (h0, w0)}  # spatial size per NCHW slot at the export size (0 = unknown/symbolic)
    for op in graph.ops:
        x = val[op.in0]
        t = op.type
        if t in (mf.OP_CONV, mf.OP_DWCONV3):
            if t == mf.OP_CONV:
                w = np.transpose(op.weights[0], (3, 2, 0, 1))  # [kh,kw,ci,co] -> [co,ci,kh,kw]
                group = 1
            else:
                w = np.transpose(op.weights[0], (2, 0, 1))[:, None]  # [3,3,c] -> [c,1,3,3]
                group = op.cin
            y = E.node("Conv", [x, E.const(w, "w"), E.const(op.weights[1], "b")],
                       {"dilations": [1, 1], "group": group, "kernel_shape": [op.kh, op.kw],
                        "pads": [op.kh // 2, op.kw // 2, op.kh // 2, op.kw // 2], "strides": [1, 1]})
            if op.relu:
                y = E.node("Relu", [y])
            hw[op.out] = hw[op.in0]
        elif t in (mf.OP_MAXPOOL, mf.OP_AVGPOOL):
            y = E.node("MaxPool" if t == mf.OP_MAXPOOL else "AveragePool", [x],
                       {"ceil_mode": 0, "kernel_shape": [op.kh, op.kw], "pads": [0, 0, 0, 0], "strides": [op.kh, op.kw]})
            hw[op.out] = (hw[op.in0][0] // op.kh, hw[op.in0][1] // op.kw)
        elif t == mf.OP_CONVT2:
            w = np.transpose(op.weights[0], (2, 3, 0, 1))  # [2,2,ci,co] -> [ci,co,2,2]
            y = E.node("ConvTranspose", [x, E.const(w, "w"), E.const(op.weights[1], "b")],
                       {"dilations": [1, 1], "group": 1, "kernel_shape": [2, 2], "pads": [0, 0, 0, 0], "strides": [2, 2]})
            hw[op.out] = (2 * hw[op.in0][0], 2 * hw[op.in0][1])
        elif t == mf.OP_PADCAT:
            (sh, sw), (uh, uw) = hw[op.in0], hw[op.in1]
            if not sh or not uh:
                raise ValueError("PADCAT needs a fixed input size (pass in_hw)")
            dy, dx = sh - uh, sw - uw
            # F.pad(x1, [dx//2, dx - dx//2, dy//2, dy - dy//2]) as torch lowers it:
            pads = E.constant_node(np.array([dx // 2, dx - dx // 2, dy // 2, dy - dy // 2], np.int64))
            ext = E.node("ConstantOfShape", [E.constant_node(np.array([4], np.int64))], {"value": np.zeros(1, np.int64)})
            cat = E.node("Concat", [pads, ext], {"axis": 0})
            r1 = E.node("Reshape", [cat, E.constant_node(np.array([-1, 2], np.int64))])
            sl = E.node("Slice", [r1, E.constant_node(np.array([-1], np.int64)), E.constant_node(np.array([INT64_MIN], np.int64)),
                                  E.constant_node(np.array([0], np.int64)), E.constant_node(np.array([-1], np.int64))])
            tr = E.node("Transpose", [sl], {"perm": [1, 0]})
            r2 = E.node("Reshape", [tr, E.constant_node(np.array([-1], np.int64))])
            pv = E.node("Cast", [r2], {"to": 7})
            padded = E.node("Pad", [val[op.in1], pv, E.constant_node(np.array(0.0, np.float32))], {"mode": "constant"})
            y = E.node("Concat", [x, padded], {"axis": 1})
            hw[op.out] = hw[op.in0]
        elif t == mf.OP_SIGMOID:
            y = E.node("Sigmoid", [x])
            hw[op.out] = hw[op.in0]
        elif t == mf.OP_TOSEQ and toseq == "squeeze":
            y = E.node("Transpose", [E.node("Squeeze", [x, E.constant_node(np.array([2], np.int64))])], {"perm": [2, 0, 1]})
        elif t == mf.OP_TOSEQ:
            # x.reshape(N, -1, W).permute(2, 0, 1)
            shp = E.node("Shape", [x])
            n = E.node("Unsqueeze", [E.node("Gather", [shp, E.constant_node(np.array(0, np.int64))], {"axis": 0}),
                                     E.constant_node(np.array([0], np.int64))])
            shp2 = E.node("Shape", [x])
            w = E.node("Unsqueeze", [E.node("Gather", [shp2, E.constant_node(np.array(3, np.int64))], {"axis": 0}),
                                     E.constant_node(np.array([0], np.int64))])
            tgt = E.node("Concat", [n, E.constant_node(np.array([-1], np.int64)), w], {"axis": 0})
            y = E.node("Transpose", [E.node("Reshape", [x, tgt])], {"perm": [2, 0, 1]})
        elif t == mf.OP_GRU:
            H = op.hidden

            def zrn(m):  # [..., 3H] in r,z,n column order -> ONNX z,r,h row blocks
                r, z, n_ = m[..., :H], m[..., H:2 * H], m[..., 2 * H:]
                return np.concatenate([z, r, n_], axis=-1)

            W = np.stack([zrn(op.weights[4 * d]).T for d in range(2)])       # [2, 3H, I]
            R = np.stack([zrn(op.weights[4 * d + 2]).T for d in range(2)])   # [2, 3H, H]
            B = np.stack([np.concatenate([zrn(op.weights[4 * d + 1]), zrn(op.weights[4 * d + 3])]) for d in range(2)])
            shp = E.node("Shape", [x])
            n = E.node("Unsqueeze", [E.node("Gather", [shp, E.constant_node(np.array(1, np.int64))], {"axis": 0}),
                                     E.constant_node(np.array([0], np.int64))])
            hshape = E.node("Concat", [E.constant_node(np.array([2], np.int64)), n, E.constant_node(np.array([H], np.int64))],
                            {"axis": 0})
            h_init = E.node("ConstantOfShape", [hshape], {"value": np.zeros(1, np.float32)})
            out = E.name("gru")
            E.nodes.append(Node("GRU", [x, E.const(W, "W"), E.const(R, "R"), E.const(B, "B"), "", h_init], [out, ""],
                                {"direction": "bidirectional", "hidden_size": H, "linear_before_reset": 1}, name=out))
            tr = E.node("Transpose", [out], {"perm": [0, 2, 1, 3]})
            y = E.node("Reshape", [tr, E.constant_node(np.array([0, 0, -1], np.int64))])
        elif t == mf.OP_LINEAR:
            y = E.node("Add", [E.node("MatMul", [x, E.const(op.weights[0], "w")]), E.const(op.weights[1], "b")])
        elif t == mf.OP_LOGSOFTMAX:
            y = E.node("LogSoftmax", [x], {"axis": 2})
        else:
            raise ValueError("cannot export op type %d" % t)
        val[op.out] = y
    out_name = val[graph.out_slot]
    E.nodes.append(Node("Identity", [out_name], ["output"], name="output_identity"))
    in_dims = ["batch", 1, graph.input_shape[2] if graph.input_shape[2] > 0 else "height",
               graph.input_shape[3] if graph.input_shape[3] > 0 else "width"]
    out_dims = ["seq", "batch", "classes"] if graph.kind == mf.KIND_RECOGNITION else ["batch", 1, "height", "width"]
    return make_model(E.nodes, E.inits, [("input", in_dims)], [("output", out_dims)], opset=opset)
