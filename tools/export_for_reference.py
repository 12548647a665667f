#!/usr/bin/env python
"""Write the seeded synthetic models as ONNX files so that the reference itself can run them
(INTEGRATION.md §5): on a machine with Rust,

    python tools/export_for_reference.py out_dir          # det.onnx rec.onnx (+ det.ocrsm rec.ocrsm, page.png)
    rten-convert out_dir/det.onnx out_dir/det.rten && rten-convert out_dir/rec.onnx out_dir/rec.rten
    ocrs --detect-model out_dir/det.rten --rec-model out_dir/rec.rten --json out_dir/page.png > ref.json
    python -m ocrs_amd.cli --detect-model out_dir/det.ocrsm --rec-model out_dir/rec.ocrsm --json out_dir/page.png > hip.json

and compare boxes and text.  The recognition head is calibrated with the CPU oracle here (no GPU needed), so
the files differ from bench.py's HIP-calibrated ones only in that bias vector."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main(out_dir):
    import models_util as M
    from ocrs_amd import synth
    from ocrs_amd import modelfile as mf
    from ocrs_amd.onnx_export import export_onnx
    from ocrs_amd.onnx_import import import_onnx
    os.makedirs(out_dir, exist_ok=True)
    for name, buf in (("det", M.detection_model_bytes()), ("rec", M.recognition_model_bytes())):
        g = mf.Graph.from_bytes(buf)
        onnx = export_onnx(g)
        assert import_onnx(onnx).to_bytes() == buf, "round trip"
        open(os.path.join(out_dir, name + ".ocrsm"), "wb").write(buf)
        open(os.path.join(out_dir, name + ".onnx"), "wb").write(onnx)
        print("%s: %d ops, %.1f MB ONNX" % (name, len(g.ops), len(onnx) / 1e6))
    from PIL import Image
    Image.fromarray(synth.synthetic_page(0, 1024, 1024, lines=80), "RGB").save(os.path.join(out_dir, "page.png"))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "export")
