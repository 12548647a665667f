#!/usr/bin/env python
"""Write the seeded synthetic models as ONNX files so that the reference itself can run them
(INTEGRATION.md §5): on a machine with Rust,

    python tools/export_for_reference.py out_dir          # det.onnx rec.onnx (+ det.ocrsm rec.ocrsm, page.png)
    rten-convert out_dir/det.onnx out_dir/det.rten && rten-convert out_dir/rec.onnx out_dir/rec.rten
    ocrs --detect-model out_dir/det.rten --rec-model out_dir/rec.rten --json out_dir/page.png > ref.json
    python -m ocrs_amd.cli --detect-model out_dir/det.ocrsm --rec-model out_dir/rec.ocrsm --json out_dir/page.png > hip.json

and compare boxes and text.  The recognition head is calibrated with the CPU oracle here (no GPU needed), so
the files differ from bench.py's HIP-calibrated ones only in that bias vector.

Round 5: closing rows f1 / (c) for a maintainer with cargo in ONE command.  The export also holds the reference's own three
images (why-rust.png, polar-bears.png, rust-book.png — the pixels the golden fixtures were made from), the per-image
detection files (det_<name>.onnx / .ocrsm: the synthetic detector's hand-set "ink" operating point differs per image),
what THIS engine's spec gives for them in the reference CLI's own JSON format (<name>.expected.json: every line's text,
word and line vertices — built from the oracle-made golden fixtures tests/golden/reference/*.npz, no GPU needed) and
check_against_reference.sh + compare_json.py:

    sh out_dir/check_against_reference.sh            # needs `ocrs` and `rten-convert` on PATH; prints OK or the first differences
"""
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
    # ---- the reference's three images: pixels, per-image detection file, expected output in the CLI's JSON format
    from ocrs_amd import TextChar, TextLine
    from ocrs_amd.output import format_json_output
    names = ("why-rust", "polar-bears", "rust-book")
    for name in names:
        g = np.load(os.path.join(ROOT, "tests", "golden", "reference", name + ".npz"))
        px = g["pixels"]
        Image.fromarray(px, "RGB").save(os.path.join(out_dir, name + ".png"))   # (rust-book: PNG of the decoded JPEG pixels)
        dbuf = M.detection_model_bytes(ink=tuple(g["ink"]))
        assert [M.digest(dbuf), M.digest(M.recognition_model_bytes())] == list(g["model_digests"]), "fixtures made with other model files"
        open(os.path.join(out_dir, "det_%s.ocrsm" % name), "wb").write(dbuf)
        open(os.path.join(out_dir, "det_%s.onnx" % name), "wb").write(export_onnx(mf.Graph.from_bytes(dbuf)))
        chars, coff = g["chars"], g["char_offsets"]
        lines = []
        for i in range(len(coff) - 1):
            cs = [TextChar(chr(int(c[0])), tuple(int(v) for v in c[1:5])) for c in chars[coff[i]:coff[i + 1]]]
            lines.append(TextLine(cs) if cs else None)
        open(os.path.join(out_dir, name + ".expected.json"), "w").write(format_json_output(name + ".png", px.shape[:2], lines))
        print("%s: %d lines, %d chars expected" % (name, sum(l is not None for l in lines), len(chars)))
    open(os.path.join(out_dir, "compare_json.py"), "w").write(COMPARE)
    open(os.path.join(out_dir, "check_against_reference.sh"), "w").write(CHECK % {"names": " ".join(names)})


COMPARE = '''#!/usr/bin/env python
"""compare_json.py expected.json got.json — the reference CLI's --json output against this engine's spec: text of every
line and word, vertices of every line and word (exact integers).  Exit 0 = identical."""
import json
import sys
a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
la, lb = a["paragraphs"][0]["lines"], b["paragraphs"][0]["lines"]
bad = []
if (a["image_width"], a["image_height"]) != (b["image_width"], b["image_height"]):
    bad.append("image size %s vs %s" % ((a["image_width"], a["image_height"]), (b["image_width"], b["image_height"])))
if len(la) != len(lb):
    bad.append("%d lines expected, %d found" % (len(la), len(lb)))
for i, (x, y) in enumerate(zip(la, lb)):
    if x["text"] != y["text"]:
        bad.append("line %d text %r vs %r" % (i, x["text"], y["text"]))
    if x["vertices"] != y["vertices"]:
        bad.append("line %d vertices %s vs %s" % (i, x["vertices"], y["vertices"]))
    for j, (w, v) in enumerate(zip(x["words"], y["words"])):
        if w != v:
            bad.append("line %d word %d %s vs %s" % (i, j, w, v))
print("OK: %d lines identical" % len(la) if not bad else "\\n".join(bad[:20] + ["... %d differences" % len(bad)]))
sys.exit(1 if bad else 0)
'''

CHECK = '''#!/bin/sh
# Runs the REFERENCE (ocrs-cli + rten) on the exported synthetic models and the reference's own three images and compares its
# JSON with what this engine's numeric spec gives (oracle-made fixtures).  Needs `ocrs` and `rten-convert` on PATH.
set -e
cd "$(dirname "$0")"
rten-convert rec.onnx rec.rten
rc=0
for n in %(names)s; do
  rten-convert det_$n.onnx det_$n.rten
  ocrs --detect-model det_$n.rten --rec-model rec.rten --json $n.png > $n.reference.json
  python compare_json.py $n.expected.json $n.reference.json || rc=1
done
exit $rc
'''


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "export")
