"""What the relaxed numerics change: the same inputs through an exact and a relaxed engine (ocrs_engine_params.numerics),
compared at the outputs the reference's contract names (BASELINE.json north_star: word / line boxes and CTC token indices
identical, detection logits and recognition log-probs within a tolerance).

Everything goes through the public API of `OcrEngine`; used by bench.py (extras.relaxed), tools/relaxed_report.py and the
`-m gpu` tests.  A "flip" is an output that differs between the two modes:
    box flips    word rects that differ (any of the 6 floats), or a differing number of words
    token flips  recognised lines whose greedy-CTC (label, position) sequence differs; label flips: lines whose LABEL sequence (the
                 decoded text) differs, with the edit distance between the two sequences
    char-box flips  characters whose decoded rect differs (lines with equal tokens only)
"""
import numpy as np

from . import DimOrder, ImageSource


def _finite_absdiff(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    both = np.isfinite(a) & np.isfinite(b)
    odd = int(np.count_nonzero(np.isfinite(a) != np.isfinite(b)))   # -inf / nan on one side only
    return (float(np.max(np.abs(a[both] - b[both]))) if both.any() else 0.0), odd


def _edit_distance(a, b):
    """Levenshtein distance between two label sequences."""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def compare_page(exact, relaxed, inp, lines=None, want_prob_map=True):
    """One prepared page (OcrInput) through both engines.  lines: use these text lines (lists of word rects) instead of
    detecting them (recognition-only inputs)."""
    out = {"words": 0, "box_flips": 0, "lines": 0, "tokens": 0, "token_flip_lines": 0, "label_flip_lines": 0, "label_edits": 0, "chars": 0, "char_box_flips": 0,
           "max_abs_dlogprob": 0.0, "nonfinite_mismatch": 0, "max_abs_dprob_map": 0.0, "flipped": []}
    if lines is None:
        we, wr = exact.detect_words(inp), relaxed.detect_words(inp)
        out["words"] = int(len(we))
        if we.shape != wr.shape:
            out["box_flips"] = int(max(len(we), len(wr)))
        else:
            out["box_flips"] = int(np.count_nonzero(np.any(we != wr, axis=1))) if len(we) else 0
        if want_prob_map:
            d, odd = _finite_absdiff(exact.detect_text_pixels(inp), relaxed.detect_text_pixels(inp))
            out["max_abs_dprob_map"] = d
            out["nonfinite_mismatch"] += odd
        lines = exact.find_text_lines(inp, we)
    out["lines"] = len(lines)
    if not len(lines):
        return out
    te, tr = exact.recognize_tokens(inp, lines), relaxed.recognize_tokens(inp, lines)
    same = []
    for i, (a, b) in enumerate(zip(te, tr)):
        out["tokens"] += len(a)
        if a != b:
            out["token_flip_lines"] += 1
            la, lb = [x[0] for x in a], [x[0] for x in b]
            if la != lb:   # the decoded TEXT differs (otherwise only the time step a token is attributed to moved)
                out["label_flip_lines"] += 1
                out["label_edits"] += _edit_distance(la, lb)
            if len(out["flipped"]) < 8:
                out["flipped"].append({"line": i, "exact": a[:40], "relaxed": b[:40]})
        same.append(a == b)
    ce, cr = exact.recognize_text(inp, lines), relaxed.recognize_text(inp, lines)
    for ok, a, b in zip(same, ce, cr):
        if not ok or a is None or b is None:
            continue
        ra, rb = [c.rect for c in a.chars()], [c.rect for c in b.chars()]
        out["chars"] += len(ra)
        out["char_box_flips"] += sum(1 for x, y in zip(ra, rb) if x != y) + abs(len(ra) - len(rb))
    le, lr = exact.recognize_logits(inp, lines), relaxed.recognize_logits(inp, lines)
    for a, b in zip(le, lr):
        if a.shape != b.shape:
            out["nonfinite_mismatch"] += 1
            continue
        d, odd = _finite_absdiff(a, b)
        out["max_abs_dlogprob"] = max(out["max_abs_dlogprob"], d)
        out["nonfinite_mismatch"] += odd
    return out


def merge(reports):
    tot = {}
    for r in reports:
        for k, v in r.items():
            if k == "flipped":
                tot.setdefault(k, [])
                tot[k] = (tot[k] + v)[:8]
            elif k.startswith("max_"):
                tot[k] = max(tot.get(k, 0.0), v)
            else:
                tot[k] = tot.get(k, 0) + v
    return tot


def compare_pixels(exact, relaxed, pages_u8_hwc, want_prob_map=True):
    """RGB u8 pages (numpy, HWC) through the whole pipeline of both engines."""
    reps = []
    for px in pages_u8_hwc:
        inp = exact.prepare_input(ImageSource.from_tensor(np.ascontiguousarray(px), DimOrder.Hwc))
        reps.append(compare_page(exact, relaxed, inp, want_prob_map=want_prob_map))
    return merge(reps)


def crops_request(engine, synth, n=2048, seed=1000):
    """BASELINE.json configs[2]: n line crops of 64 x 256 stacked into one tall grey page, one word rect per crop."""
    crops = synth.synthetic_line_crops(seed, n=n)
    page = (crops.reshape(1, n * 64, 256) + 0.5).astype(np.float32)
    inp = engine.prepare_input(ImageSource.from_tensor(page, DimOrder.Chw))
    rects = np.zeros((n, 6), np.float32)
    rects[:, 0] = 128.0
    rects[:, 1] = np.arange(n) * 64.0 + 32.0
    rects[:, 2], rects[:, 3] = 0.0, 1.0
    rects[:, 4], rects[:, 5] = 256.0, 64.0
    return inp, [rects[i:i + 1] for i in range(n)]
