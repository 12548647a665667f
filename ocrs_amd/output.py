"""Output formats of ocrs-cli (ocrs-cli/src/output.rs): plain text and the
HierText-like JSON (`paragraphs[].lines[].{text, vertices, words[]}`)."""
import json
import math

from . import rotated_rect_corners


def _rround(v):  # f32::round — half away from zero
    return int(math.copysign(math.floor(abs(float(v)) + 0.5), float(v)))


def rounded_vertex_coords(rect6):
    """output.rs:24-27."""
    return [[_rround(x), _rround(y)] for x, y in rotated_rect_corners(rect6)]


def ocr_json(input_path, input_hw, text_lines):
    """output.rs:34-76.  text_lines: list of TextLine | None."""
    line_items = []
    for line in text_lines:
        if line is None:
            continue
        words = [{"text": str(w), "vertices": rounded_vertex_coords(w.rotated_rect())} for w in line.words()]
        line_items.append({"text": str(line), "words": words, "vertices": rounded_vertex_coords(line.rotated_rect())})
    height, width = input_hw
    return {"url": input_path, "image_width": width, "image_height": height, "paragraphs": [{"lines": line_items}]}


def format_json_output(input_path, input_hw, text_lines):
    """output.rs:98-101 (serde_json::to_string_pretty).  serde_json without `preserve_order` keeps `json!` maps in a
    BTreeMap, so the reference emits every object's keys in alphabetical order (ocrs-cli/test-data/
    format-json-expected.json: image_height, image_width, paragraphs, url / text, vertices, words): sort_keys."""
    return json.dumps(ocr_json(input_path, input_hw, text_lines), indent=2, ensure_ascii=False, sort_keys=True)


def format_text_output(text_lines):
    """output.rs:88-95."""
    return "\n".join(str(l) for l in text_lines if l is not None)
