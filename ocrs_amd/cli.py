"""`python -m ocrs_amd.cli image.png` — the call sequence of ocrs-cli
(ocrs-cli/src/main.rs:366-497) on the MI355X engine: load models, decode the
image to RGB u8 HWC (main.rs:312-323), prepare_input -> detect_words ->
find_text_lines -> recognize_text, print text or JSON.

Differences that are forced by the environment: models are `.ocrsm` files
(--detect-model / --rec-model; there is no network to download the default
`.rten` files from, main.rs:305-309) — with neither flag the seeded synthetic
models of ocrs_amd.models are used; the annotated-PNG output (-p) is not provided.
The debug dumps (--text-map / --text-mask / --text-line-images, main.rs:422-444)
write the same greyscale PNGs as the reference: (x.clamp(0,1) * 255) as u8
(main.rs:44-51).
"""
import os
import argparse
import sys

import numpy as np


def load_image(path):
    """main.rs:312-323: image::open(..).into_rgb8() -> [H, W, 3] u8."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))


def write_image(path, chw_or_hw):
    """main.rs:21-51: float tensor in [0, 1] -> 8-bit greyscale PNG, `(x.clamp(0., 1.) * 255.0) as u8` (truncating)."""
    from PIL import Image
    a = np.asarray(chw_or_hw, np.float32)
    a = a.reshape(a.shape[-2], a.shape[-1])
    Image.fromarray((np.clip(a, np.float32(0.0), np.float32(1.0)) * np.float32(255.0)).astype(np.uint8), "L").save(path)


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
    ap.add_argument("--text-map", action="store_true", help="write text-map.png (detect_text_pixels)")
    ap.add_argument("--text-mask", action="store_true", help="write text-mask.png (text map > detection threshold)")
    ap.add_argument("--numerics", choices=("exact", "relaxed", "reduced"), default="exact",
                    help="ocrs_engine_params.numerics (no reference counterpart): exact = bits of the CPU restatement (default); "
                         "relaxed / reduced = faster arithmetic whose outputs are expected, not guaranteed, to match (DESIGN.md 4.4)")
    ap.add_argument("--text-line-images", action="store_true",
                    help="write lines/line-N.png: the pre-processed recognition input of every text line")
    args = ap.parse_args(argv)

    from . import DecodeMethod, DimOrder, ImageSource, Model, OcrEngine, models, output
    from ._lib import OcrsError
    det = Model.load_file(args.detect_model) if args.detect_model else Model.load_bytes(models.synthetic_detection_bytes())
    rec = Model.load_file(args.rec_model) if args.rec_model else Model.load_bytes(models.synthetic_recognition_bytes())
    engine = OcrEngine(detection_model=det, recognition_model=rec, debug=args.debug, alphabet=args.alphabet,
                       allowed_chars=args.allowed_chars, numerics=args.numerics,
                       decode_method=DecodeMethod.BeamSearch(100) if args.beam else DecodeMethod.Greedy)
    # JPEG files: Huffman decoding here, everything per-sample on the GPU (include/ocrs_amd.h "JPEG hand-off"); flavours
    # the hand-off does not cover, and every other format, are decoded on the host as the reference does (main.rs:312-323)
    inp, shape_hw = None, None
    with open(args.image, "rb") as f:
        head = f.read(2)
        if head == b"\xff\xd8" and not os.environ.get("OCRS_CLI_HOST_DECODE"):
            data = head + f.read()
            try:
                inp, coef_bytes = engine.prepare_input_jpeg(data)
                shape_hw = inp.shape[-2:]
                if args.debug:
                    print("JPEG hand-off: %d bytes of coefficients to the GPU for %dx%d pixels" % (coef_bytes, shape_hw[1], shape_hw[0]))
            except OcrsError as e:
                if e.status != 6:   # OCRS_ERR_IMAGE_SOURCE = a flavour to decode on the host
                    raise
    if inp is None:
        img = load_image(args.image)
        shape_hw = img.shape[:2]
        inp = engine.prepare_input(ImageSource.from_tensor(img, DimOrder.Hwc))
    if args.text_map or args.text_mask:
        tm = engine.detect_text_pixels(inp)
        if args.text_map:
            write_image("text-map.png", tm)
        if args.text_mask:
            write_image("text-mask.png", (tm > np.float32(engine.detection_threshold())).astype(np.float32))
    words = engine.detect_words(inp)
    lines = engine.find_text_lines(inp, words)
    if args.text_line_images:  # main.rs:66-86
        os.makedirs("lines", exist_ok=True)
        for i, line in enumerate(lines):
            write_image("lines/line-%d.png" % i, engine.prepare_recognition_input(inp, line) + np.float32(0.5))
    texts = engine.recognize_text(inp, lines)
    if args.json:
        content = output.format_json_output(args.image, tuple(shape_hw), texts)
    else:
        content = output.format_text_output(texts)
    if args.output:
        with open(args.output, "w", encoding="utf-8") as f:
            f.write(content)
    else:
        print(content)
    if args.debug:
        print("Found %d words, %d lines in image of size %dx%d" % (len(words), len(lines), shape_hw[1], shape_hw[0]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
