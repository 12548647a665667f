#!/usr/bin/env python
"""What one step of the persistent recurrence kernels costs as a function of the row tiles a wave carries.

    python tools/gru_round_probe.py [--width 1024]
    python -m ocrs_amd.build --variant probe kernels_gru_split.hip OCRS_GRU_PROBE
    OCRS_AMD_LIB=$PWD/ocrs_amd/libocrs_amd.probe.so python tools/gru_round_probe.py --once relaxed 128

Requests of n x 16 lines of EQUAL length (crops of 64 x width: width / 4 steps), n = 1 .. 128 row tiles, through the
recognition model alone in every numerics mode.  With equal lengths the deal gives every wave the same number of tiles
(ceil(n / 32) once the 8 clusters per direction are in use), so   layer time / steps   is the round time for that many tiles:
the constants behind gru_assign_tiles' cost model (kernels_gru.hip) and their counterparts for kernels_gru_split.hip.
--once <mode> <tiles>: one such request and nothing else — with the OCRS_GRU_PROBE build of kernels_gru_split.hip its waves print
where an item's cycles go (matrix chain, gates, stores, waiting for the state).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import models_util as M  # noqa: E402
from ocrs_amd import DimOrder, ImageSource, Model, OcrEngine, synth  # noqa: E402


def request(engine, n_lines, width):
    crops = synth.synthetic_line_crops(7, n=min(n_lines, 64), width=width)
    crops = crops[np.arange(n_lines) % len(crops)]
    page = (crops.reshape(1, n_lines * 64, width) + 0.5).astype(np.float32)
    inp = engine.prepare_input(ImageSource.from_tensor(page, DimOrder.Chw))
    rects = np.zeros((n_lines, 6), np.float32)
    rects[:, 0] = width / 2
    rects[:, 1] = np.arange(n_lines) * 64.0 + 32.0
    rects[:, 3] = 1.0
    rects[:, 4], rects[:, 5] = width, 64.0
    return inp, [rects[i:i + 1] for i in range(n_lines)]


def main():
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 1024
    rec = Model.load_bytes(M.recognition_model_bytes())
    if "--once" in sys.argv:
        mode, tiles = sys.argv[sys.argv.index("--once") + 1], int(sys.argv[sys.argv.index("--once") + 2])
        eng = OcrEngine(recognition_model=rec, numerics=mode)
        inp, lines = request(eng, tiles * 16, width)
        eng.recognize_tokens(inp, lines)
        print("====", mode, tiles, flush=True)
        eng.recognize_tokens(inp, lines)
        return
    out = {"width": width, "steps": None, "us_per_step_and_layer": {}}
    for mode in ("exact", "relaxed", "reduced"):
        eng = OcrEngine(recognition_model=rec, numerics=mode, options={"gru_gates": 0})   # the general kernels only
        row = {}
        for tiles in (1, 8, 32, 64, 96, 128):
            inp, lines = request(eng, tiles * 16, width)
            toks = eng.recognize_tokens(inp, lines)
            eng.enable_timing(2)
            eng.kernel_stats(reset=True)
            for _ in range(3):
                eng.recognize_tokens(inp, lines)
            ks = eng.kernel_stats(reset=True)["gemm_gru_hidden_mfma"]
            eng.enable_timing(0)
            steps = len(eng.recognize_logits(inp, lines[:1])[0])
            out["steps"] = steps
            row[tiles] = round(1e3 * ks["ms"] / ks["launches"] / steps, 3)
            assert len(toks) == tiles * 16
        out["us_per_step_and_layer"][mode] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
