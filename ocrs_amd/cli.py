"""`python -m ocrs_amd.cli image.png` — the call sequence of ocrs-cli
(ocrs-cli/src/main.rs:366-497) on the MI355X engine: load models, decode the
image to RGB u8 HWC (main.rs:312-323), prepare_input -> detect_words ->
find_text_lines -> recognize_text, print text or JSON.

Differences that are forced by the environment: models are `.ocrsm` files
(--detect-model / --rec-model; there is no network to download the default
`.rten` files from, main.rs:305-309) — with neither flag the seeded synthetic
models of ocrs_amd.models are used; PNG annotation output is not provided.
"""
import argparse
import sys

import numpy as np


def load_image(path):
    """main.rs:312-323: image::open(..).into_rgb8() -> [H, W, 3] u8."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="ocrs_amd", description="Extract text from an image (MI355X engine).")
    ap.add_argument("image")
    ap.add_argument("--detect-model")
    ap.add_argument("--rec-model")
    ap.add_argument("--alphabet")
    ap.add_argument("--allowed-chars")
    ap.add_argument("--beam", action="store_true", help="beam search decoding (width 100, main.rs:403-404)")
    ap.add_argument("-j", "--json", action="store_true")
    ap.add_argument("-o", "--output")
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("--text-map", action="store_true", help="write text-map.npy (detect_text_pixels)")
    ap.add_argument("--text-mask", action="store_true", help="write text-mask.npy")
    args = ap.parse_args(argv)

    from . import DecodeMethod, DimOrder, ImageSource, Model, OcrEngine, models, output
    det = Model.load_file(args.detect_model) if args.detect_model else Model.load_bytes(models.synthetic_detection_bytes())
    rec = Model.load_file(args.rec_model) if args.rec_model else Model.load_bytes(models.synthetic_recognition_bytes())
    engine = OcrEngine(detection_model=det, recognition_model=rec, debug=args.debug, alphabet=args.alphabet,
                       allowed_chars=args.allowed_chars,
                       decode_method=DecodeMethod.BeamSearch(100) if args.beam else DecodeMethod.Greedy)
    img = load_image(args.image)
    inp = engine.prepare_input(ImageSource.from_tensor(img, DimOrder.Hwc))
    if args.text_map or args.text_mask:
        tm = engine.detect_text_pixels(inp)
        if args.text_map:
            np.save("text-map.npy", tm)
        if args.text_mask:
            np.save("text-mask.npy", (tm > np.float32(engine.detection_threshold())).astype(np.uint8))
    words = engine.detect_words(inp)
    lines = engine.find_text_lines(inp, words)
    texts = engine.recognize_text(inp, lines)
    if args.json:
        content = output.format_json_output(args.image, img.shape[:2], texts)
    else:
        content = output.format_text_output(texts)
    if args.output:
        with open(args.output, "w", encoding="utf-8") as f:
            f.write(content)
    else:
        print(content)
    if args.debug:
        print("Found %d words, %d lines in image of size %dx%d" % (len(words), len(lines), img.shape[1], img.shape[0]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
