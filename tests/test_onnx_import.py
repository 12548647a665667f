"""ONNX import.  The real model files are unobtainable offline, so
  (1) the exporter is checked against the ONNX operator semantics with an independent
      evaluator (torch functional ops as the executable spec, ONNX GRU equations in numpy),
  (2) importer(exporter(graph)) must reproduce the container byte for byte,
  (3) unsupported constructs must be refused with the node named.
CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ocrs_amd import modelfile as mf
from ocrs_amd import onnx_pb as pb
from ocrs_amd.onnx_export import export_onnx
from ocrs_amd.onnx_import import OnnxImportError, import_onnx
from oracle.nn import OracleGraph


def small_detection():
    return mf.build_detection(in_hw=(100, 76), depths=(4, 8, 8, 16), seed=5)


def small_recognition():
    return mf.build_recognition(n_classes=11, in_h=32, seed=6, hidden=8, chans=(4, 8, 8, 8, 8, 8))


# ---------------------------------------------------------------------------------------
# Independent evaluator of the exported ONNX graph (ONNX operator spec semantics).
# ---------------------------------------------------------------------------------------
def onnx_gru(X, W, R, B, H):
    """ONNX GRU, bidirectional, linear_before_reset=1 (onnx/docs/Operators.md#GRU), float64."""
    T, N, _ = X.shape
    Y = np.zeros((T, 2, N, H))
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    for d in range(2):
        Wz, Wr, Wh = W[d][:H], W[d][H:2 * H], W[d][2 * H:]
        Rz, Rr, Rh = R[d][:H], R[d][H:2 * H], R[d][2 * H:]
        Wbz, Wbr, Wbh, Rbz, Rbr, Rbh = [B[d][i * H:(i + 1) * H] for i in range(6)]
        h = np.zeros((N, H))
        for t in (range(T) if d == 0 else reversed(range(T))):
            x = X[t]
            z = sig(x @ Wz.T + h @ Rz.T + Wbz + Rbz)
            r = sig(x @ Wr.T + h @ Rr.T + Wbr + Rbr)
            hh = np.tanh(x @ Wh.T + r * (h @ Rh.T + Rbh) + Wbh)
            h = (1 - z) * hh + z * h
            Y[t, d] = h
    return Y


def eval_onnx(data, x):
    m = pb.parse_model(data)
    env = {k: torch.from_numpy(np.asarray(v)) for k, v in m.initializers.items()}
    env[m.inputs[0][0]] = torch.from_numpy(x).double()
    for n in m.nodes:
        i = [env[k] if k else None for k in n.inputs]
        a = n.attrs
        if n.op == "Constant":
            o = torch.from_numpy(np.asarray(a["value"]))
        elif n.op == "Conv":
            o = F.conv2d(i[0], i[1].double(), i[2].double(), stride=a["strides"], padding=a["pads"][:2], groups=a["group"])
        elif n.op == "Relu":
            o = F.relu(i[0])
        elif n.op == "MaxPool":
            o = F.max_pool2d(i[0], a["kernel_shape"], a["strides"])
        elif n.op == "AveragePool":
            o = F.avg_pool2d(i[0], a["kernel_shape"], a["strides"])
        elif n.op == "ConvTranspose":
            o = F.conv_transpose2d(i[0], i[1].double(), i[2].double(), stride=a["strides"])
        elif n.op == "ConstantOfShape":
            v = np.asarray(a["value"]).reshape(-1)[0]
            o = torch.full([int(d) for d in i[0]], v, dtype=torch.from_numpy(np.asarray(a["value"])).dtype)
        elif n.op == "Concat":
            o = torch.cat([t.reshape(-1) if t.dim() == 0 else t for t in i], dim=a["axis"])
        elif n.op == "Reshape":
            shp = [int(i[0].shape[k]) if int(v) == 0 else int(v) for k, v in enumerate(i[1])]
            o = i[0].reshape(shp)
        elif n.op == "Slice":
            assert int(i[4][0]) == -1 and int(i[3][0]) == 0 and int(i[1][0]) == -1
            o = torch.flip(i[0], dims=[0])
        elif n.op == "Transpose":
            o = i[0].permute(a["perm"])
        elif n.op == "Cast":
            o = i[0].to(torch.int64)
        elif n.op == "Pad":
            p = [int(v) for v in i[1]]
            o = F.pad(i[0], [p[3], p[7], p[2], p[6]])
        elif n.op == "Sigmoid":
            o = torch.sigmoid(i[0])
        elif n.op == "Shape":
            o = torch.tensor(list(i[0].shape), dtype=torch.int64)
        elif n.op == "Gather":
            o = i[0][int(i[1])]
        elif n.op == "Unsqueeze":
            o = i[0].reshape(1)
        elif n.op == "GRU":
            assert a["linear_before_reset"] == 1 and a["direction"] in (b"bidirectional", "bidirectional")
            assert float(i[5].abs().sum()) == 0.0
            o = torch.from_numpy(onnx_gru(i[0].numpy(), i[1].double().numpy(), i[2].double().numpy(),
                                          i[3].double().numpy(), a["hidden_size"]))
        elif n.op == "MatMul":
            o = i[0] @ i[1].double()
        elif n.op == "Add":
            o = i[0] + i[1].double()
        elif n.op == "LogSoftmax":
            o = F.log_softmax(i[0], dim=a["axis"])
        elif n.op == "Identity":
            o = i[0]
        else:
            raise AssertionError("evaluator: op %s" % n.op)
        env[n.outputs[0]] = o
    return env[m.outputs[0][0]].numpy()


# ---------------------------------------------------------------------------------------
def test_protobuf_codec_roundtrip():
    nodes = [pb.Node("Conv", ["x", "w", ""], ["y"], {"pads": [1, 1, 1, 1], "group": 3, "alpha": 0.5, "mode": "constant",
                                                      "value": np.arange(6, dtype=np.int64).reshape(2, 3), "empty": []}, name="c0")]
    inits = {"w": np.arange(24, dtype=np.float32).reshape(2, 3, 2, 2), "neg": np.array([-1, -(2 ** 62)], np.int64)}
    data = pb.make_model(nodes, inits, [("x", ["batch", 1, 64, "width"])], [("y", ["seq", "batch", 97])])
    m = pb.parse_model(data)
    assert m.opset == 17 and m.inputs == [("x", [-1, 1, 64, -1])] and m.outputs == [("y", [-1, -1, 97])]
    n = m.nodes[0]
    assert (n.op, n.inputs, n.outputs, n.name) == ("Conv", ["x", "w", ""], ["y"], "c0")
    assert n.attrs["pads"] == [1, 1, 1, 1] and n.attrs["group"] == 3 and n.attrs["alpha"] == 0.5
    assert n.attrs["mode"] == b"constant" and n.attrs["empty"] == []
    assert np.array_equal(n.attrs["value"], np.arange(6).reshape(2, 3))
    assert np.array_equal(m.initializers["w"], inits["w"]) and np.array_equal(m.initializers["neg"], inits["neg"])


def test_exported_detection_matches_onnx_semantics():
    g = small_detection()
    x = np.random.default_rng(0).uniform(-0.5, 0.5, (2, 1, 100, 76)).astype(np.float32)
    ref = eval_onnx(export_onnx(g), x)
    got = OracleGraph(g.to_bytes()).run_exact(x)
    assert ref.shape == got.shape == (2, 1, 100, 76)
    assert np.abs(ref - got).max() < 2e-5  # float64 ONNX evaluation vs the fp32 executor spec


def test_exported_recognition_matches_onnx_semantics():
    g = small_recognition()
    x = np.random.default_rng(1).uniform(-0.5, 0.5, (3, 1, 32, 48)).astype(np.float32)
    ref = eval_onnx(export_onnx(g), x)
    got = OracleGraph(g.to_bytes()).run_exact(x)
    assert ref.shape == got.shape == (12, 3, 11)
    assert np.abs(ref - got).max() < 5e-5


@pytest.mark.parametrize("make", [small_detection, small_recognition,
                                  lambda: mf.build_detection(in_hw=(800, 600), depths=(8, 16, 32, 32, 64, 128, 256)),
                                  lambda: mf.build_recognition()])
def test_import_of_export_is_identity(make):
    g = make()
    back = import_onnx(export_onnx(g))
    assert back.kind == g.kind and back.input_shape == g.input_shape
    assert back.to_bytes() == g.to_bytes()


def test_import_squeeze_idiom_and_evaluator_agree():
    g = small_recognition()
    data = export_onnx(g, toseq="squeeze")
    assert import_onnx(data).to_bytes() == g.to_bytes()


def test_imported_model_runs_in_oracle_executor():
    g = small_recognition()
    back = import_onnx(export_onnx(g))
    x = np.random.default_rng(2).uniform(-0.5, 0.5, (2, 1, 32, 64)).astype(np.float32)
    assert np.array_equal(OracleGraph(back.to_bytes()).run_exact(x), OracleGraph(g.to_bytes()).run_exact(x))


def _mutate(data, fn):
    m = pb.parse_model(data)
    fn(m)
    return pb.make_model(m.nodes, m.initializers, [(n, [d if d > 0 else "d%d" % i for i, d in enumerate(dims)]) for n, dims in m.inputs],
                         [(n, ["o%d" % i for i, _ in enumerate(dims)]) for n, dims in m.outputs])


def test_importer_refuses_what_the_executor_cannot_run():
    det, rec = export_onnx(small_detection()), export_onnx(small_recognition())

    def stride2(m):
        next(n for n in m.nodes if n.op == "Conv").attrs["strides"] = [2, 2]
    with pytest.raises(OnnxImportError, match="Conv node .*stride"):
        import_onnx(_mutate(det, stride2))

    def lbr0(m):
        next(n for n in m.nodes if n.op == "GRU").attrs["linear_before_reset"] = 0
    with pytest.raises(OnnxImportError, match="GRU node .*linear_before_reset"):
        import_onnx(_mutate(rec, lbr0))

    def softmax(m):
        next(n for n in m.nodes if n.op == "LogSoftmax").op = "Softmax"
    with pytest.raises(OnnxImportError, match="Softmax node .*not supported"):
        import_onnx(_mutate(rec, softmax))

    def offcentre(m):  # pad everything on the top/left instead of centred
        c = [n for n in m.nodes if n.op == "Constant" and np.asarray(n.attrs["value"]).shape == (4,)][0]
        v = np.asarray(c.attrs["value"]).copy()
        c.attrs["value"] = np.array([v[0] + v[1], 0, v[2] + v[3], 0], np.int64)
    m_det = _mutate(det, offcentre)
    if m_det != det:
        with pytest.raises(OnnxImportError, match="Concat node .*centred"):
            import_onnx(m_det)

    def forward_only(m):
        next(n for n in m.nodes if n.op == "GRU").attrs["direction"] = "forward"
    with pytest.raises(OnnxImportError, match="GRU node .*bidirectional"):
        import_onnx(_mutate(rec, forward_only))


def test_export_for_reference_tool_runs_end_to_end(tmp_path):
    """tools/export_for_reference.py: the artefacts a machine WITH the Rust toolchain needs to
    run the reference on the same synthetic models — det/rec as .onnx (round trip byte-identical to the .ocrsm) and
    the bench's page 0 as PNG — are produced here."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("export_for_reference", os.path.join(root, "tools", "export_for_reference.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(str(tmp_path))
    names = sorted(p.name for p in tmp_path.iterdir())
    refs = ("polar-bears", "rust-book", "why-rust")
    assert names == sorted(["det.ocrsm", "det.onnx", "page.png", "rec.ocrsm", "rec.onnx", "check_against_reference.sh", "compare_json.py"] +
                           [f % n for n in refs for f in ("%s.png", "%s.expected.json", "det_%s.onnx", "det_%s.ocrsm")])
    assert (tmp_path / "rec.onnx").stat().st_size > 9_000_000 and (tmp_path / "page.png").stat().st_size > 100_000
    # round 5: what this engine's spec gives for the reference's own images, in the reference CLI's JSON format, and the
    # comparison script a maintainer with cargo runs against `ocrs --json`
    import json
    import subprocess
    import sys
    exp = json.load(open(tmp_path / "why-rust.expected.json"))
    g = np.load(os.path.join(root, "tests", "golden", "reference", "why-rust.npz"))
    lines = exp["paragraphs"][0]["lines"]
    assert (exp["image_height"], exp["image_width"]) == g["pixels"].shape[:2]
    assert "\n".join(l["text"] for l in lines) == str(g["text"][0])
    assert all(len(l["vertices"]) == 4 and all(len(w["vertices"]) == 4 for w in l["words"]) for l in lines)
    same = subprocess.run([sys.executable, str(tmp_path / "compare_json.py"), str(tmp_path / "why-rust.expected.json"),
                           str(tmp_path / "why-rust.expected.json")], capture_output=True, text=True)
    assert same.returncode == 0 and "OK: %d lines identical" % len(lines) in same.stdout
    lines[3]["text"] += "x"
    lines[5]["words"][0]["vertices"][0][0] += 1
    json.dump(exp, open(tmp_path / "changed.json", "w"))
    diff = subprocess.run([sys.executable, str(tmp_path / "compare_json.py"), str(tmp_path / "why-rust.expected.json"),
                           str(tmp_path / "changed.json")], capture_output=True, text=True)
    assert diff.returncode == 1 and "line 3 text" in diff.stdout and "line 5 word 0" in diff.stdout
