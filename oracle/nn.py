"""Oracle executor for `*.ocrsm` fixed-graph model files (the `Model::run`
side of ocrs/src/model.rs:33-40).  TEST INFRASTRUCTURE ONLY.

Two back-ends over the same graph:
  * "exact"  — the C restatement in oracle/csrc/ocrs_oracle.c: fp32 fmaf chains
               in the canonical order of DESIGN.md §4.  The HIP executor must
               match this bit for bit.
  * "torch"  — PyTorch-CPU fp32 (`conv2d`, `conv_transpose2d`, `max_pool2d`,
               `avg_pool2d`, `GRU` math, `log_softmax`, `sigmoid`): the
               executable ONNX-operator spec, used (a) to show the exact chain
               order is a valid fp32 evaluation (tolerance in the tests) and
               (b) as the multi-threaded CPU baseline in bench.py (what RTen's
               CPU path would be doing: SURVEY.md §8(d) "CPU baseline").

Has its own reader for the container so the checker does not depend on the
product's Python package.
"""
import struct

import numpy as np

from . import clib

_HDR = struct.Struct("<8sII4qIIIIQ")
_OP = struct.Struct("<I9iII16Q")
(OP_CONV, OP_DWCONV3, OP_MAXPOOL, OP_AVGPOOL, OP_CONVT2, OP_PADCAT, OP_SIGMOID, OP_TOSEQ, OP_GRU, OP_LINEAR,
 OP_LOGSOFTMAX) = range(11)


class OracleGraph:
    def __init__(self, buf):
        magic, version, kind, n, c, h, w, n_ops, n_slots, out_slot, _, blob_floats = _HDR.unpack_from(buf, 0)
        if magic != b"OCRSMDL1" or version != 1:
            raise ValueError("not an OCRSMDL1 model file")
        self.kind = kind
        self.input_shape_raw = [n, c, h, w]
        self.out_slot = out_slot
        blob = np.frombuffer(buf, dtype="<f4", count=blob_floats, offset=_HDR.size + n_ops * _OP.size)
        self.ops = []
        for i in range(n_ops):
            f = _OP.unpack_from(buf, _HDR.size + i * _OP.size)
            d = dict(zip(("type", "in0", "in1", "out", "relu", "kh", "kw", "cin", "cout", "hidden", "n_w"), f[:11]))
            refs = f[12:]
            d["w"] = [np.array(blob[refs[2 * j]:refs[2 * j] + refs[2 * j + 1]]) for j in range(d["n_w"])]
            self.ops.append(d)
        self._torch_cache = None

    @staticmethod
    def load(path):
        with open(path, "rb") as fh:
            return OracleGraph(fh.read())

    def input_shape(self):
        """model.rs:20-31 — None marks a symbolic dimension."""
        return [None if v < 0 else int(v) for v in self.input_shape_raw]

    # ------------------------------------------------------------ exact back-end
    def run_exact(self, nchw, return_slots=False):
        x = np.ascontiguousarray(nchw, np.float32)
        n, c, h, w = x.shape
        assert c == 1
        slots = {0: x.reshape(n, h, w, 1)}
        for op in self.ops:
            t, a = op["type"], slots[op["in0"]]
            W = op["w"]
            if t == OP_CONV:
                y = clib.conv2d(a, W[0].reshape(op["kh"], op["kw"], op["cin"], op["cout"]), W[1], op["relu"])
            elif t == OP_DWCONV3:
                y = clib.dwconv3x3(a, W[0].reshape(3, 3, op["cin"]), W[1], op["relu"])
            elif t == OP_MAXPOOL:
                y = clib.maxpool(a, op["kh"], op["kw"])
            elif t == OP_AVGPOOL:
                y = clib.avgpool(a, op["kh"], op["kw"])
            elif t == OP_CONVT2:
                y = clib.convt2x2(a, W[0].reshape(2, 2, op["cin"], op["cout"]), W[1])
            elif t == OP_PADCAT:
                y = clib.padcat(a, slots[op["in1"]])
            elif t == OP_SIGMOID:
                y = clib.sigmoid(a)
            elif t == OP_TOSEQ:
                assert a.shape[1] == 1
                y = np.ascontiguousarray(a[:, 0].transpose(1, 0, 2))  # [N,W,C] -> [T,N,C]
            elif t == OP_GRU:
                hd, i = op["hidden"], op["cin"]
                ws = []
                for d in range(2):
                    wi, bi, wh, bh = W[4 * d:4 * d + 4]
                    ws += [wi.reshape(i, 3 * hd), bi, wh.reshape(hd, 3 * hd), bh]
                y = clib.gru_bidir(a, ws)
            elif t == OP_LINEAR:
                y = clib.linear(a, W[0].reshape(op["cin"], op["cout"]), W[1])
            elif t == OP_LOGSOFTMAX:
                y = clib.log_softmax(a)
            else:
                raise ValueError("bad op %d" % t)
            slots[op["out"]] = y
        out = slots[self.out_slot]
        if self.kind == 0:  # detection: NHWC -> NCHW
            out = np.ascontiguousarray(out.transpose(0, 3, 1, 2))
        return (out, slots) if return_slots else out

    # ------------------------------------------------------------ torch back-end
    def _torch_weights(self):
        import torch
        if self._torch_cache is None:
            cache = []
            for op in self.ops:
                t, W = op["type"], op["w"]
                if t == OP_CONV:
                    w = torch.from_numpy(W[0].reshape(op["kh"], op["kw"], op["cin"], op["cout"]).transpose(3, 2, 0, 1).copy())
                    cache.append((w, torch.from_numpy(W[1].copy())))
                elif t == OP_DWCONV3:
                    w = torch.from_numpy(W[0].reshape(3, 3, op["cin"]).transpose(2, 0, 1)[:, None].copy())
                    cache.append((w, torch.from_numpy(W[1].copy())))
                elif t == OP_CONVT2:
                    w = torch.from_numpy(W[0].reshape(2, 2, op["cin"], op["cout"]).transpose(2, 3, 0, 1).copy())
                    cache.append((w, torch.from_numpy(W[1].copy())))
                elif t in (OP_LINEAR, OP_GRU):
                    cache.append([torch.from_numpy(a.copy()) for a in W])
                else:
                    cache.append(None)
            self._torch_cache = cache
        return self._torch_cache

    def run_torch(self, nchw):
        import torch
        import torch.nn.functional as F
        tw = self._torch_weights()
        with torch.no_grad():
            slots = {0: torch.from_numpy(np.ascontiguousarray(nchw, np.float32))}
            for op, cw in zip(self.ops, tw):
                t, a = op["type"], slots[op["in0"]]
                if t == OP_CONV:
                    y = F.conv2d(a, cw[0], cw[1], padding=(op["kh"] // 2, op["kw"] // 2))
                    if op["relu"]:
                        y = F.relu(y)
                elif t == OP_DWCONV3:
                    y = F.conv2d(a, cw[0], cw[1], padding=1, groups=op["cin"])
                    if op["relu"]:
                        y = F.relu(y)
                elif t == OP_MAXPOOL:
                    y = F.max_pool2d(a, (op["kh"], op["kw"]))
                elif t == OP_AVGPOOL:
                    y = F.avg_pool2d(a, (op["kh"], op["kw"]))
                elif t == OP_CONVT2:
                    y = F.conv_transpose2d(a, cw[0], cw[1], stride=2)
                elif t == OP_PADCAT:
                    b = slots[op["in1"]]
                    dy, dx = a.shape[2] - b.shape[2], a.shape[3] - b.shape[3]
                    b = F.pad(b, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
                    y = torch.cat([a, b], dim=1)
                elif t == OP_SIGMOID:
                    y = torch.sigmoid(a)
                elif t == OP_TOSEQ:
                    y = a[:, :, 0, :].permute(2, 0, 1).contiguous()  # [N,C,1,W] -> [T,N,C]
                elif t == OP_GRU:
                    # ATen's native bidirectional GRU (same gate order r|z|n and the same candidate formula
                    # n = tanh(gx_n + r * (h Wh_n + bh_n)) as the graph's GRU): the time loop runs in C++, which is
                    # what a native runtime such as rten does — a Python loop of 600 small matmuls per chunk would
                    # make this baseline an order of magnitude slower than the reference's CPU path.
                    hd = op["hidden"]
                    flat = []
                    for d in range(2):
                        wi, bi, wh, bh = cw[4 * d:4 * d + 4]
                        flat += [wi.reshape(op["cin"], 3 * hd).t().contiguous(), wh.reshape(hd, 3 * hd).t().contiguous(),
                                 bi.reshape(-1).contiguous(), bh.reshape(-1).contiguous()]
                    h0 = torch.zeros(2, a.shape[1], hd)
                    y, _ = torch._VF.gru(a, h0, flat, True, 1, 0.0, False, True, False)
                elif t == OP_LINEAR:
                    y = a @ cw[0].reshape(op["cin"], op["cout"]) + cw[1]
                elif t == OP_LOGSOFTMAX:
                    y = F.log_softmax(a, dim=-1)
                slots[op["out"]] = y
            return slots[self.out_slot].numpy()


class OracleModel:
    """`impl Model for ...` (model.rs:19-41) over an OracleGraph."""

    def __init__(self, graph, backend="exact"):
        self.graph = graph
        self.backend = backend

    def input_shape(self):
        return self.graph.input_shape()

    def run(self, nchw):
        return self.graph.run_exact(nchw) if self.backend == "exact" else self.graph.run_torch(nchw)
