"""Fixtures lifted from the reference's own tests (no model weights needed)."""
import numpy as np


def gen_rect_grid(top_left_yx, grid_shape, rect_size, gap_size):
    """ocrs/src/test_util.rs:7-28 -> list of (top, left, bottom, right)."""
    y0, x0 = top_left_yx
    rows, cols = grid_shape
    rh, rw = rect_size
    gh, gw = gap_size
    out = []
    for r in range(rows):
        for c in range(cols):
            top = y0 + r * (rh + gh)
            left = x0 + c * (rw + gw)
            out.append((top, left, top + rh, left + rw))
    return out


def union_rects(rects):
    """ocrs/src/test_util.rs:31-35."""
    if not rects:
        return None
    a = np.array(rects)
    return (a[:, 0].min(), a[:, 1].min(), a[:, 2].max(), a[:, 3].max())


def gen_test_image(n_words):
    """ocrs/src/lib.rs:319-333: CHW f32 [3,100,200], white 20x50 rects at top=30."""
    img = np.zeros((3, 100, 200), np.float32)
    for w in range(n_words):
        img[:, 30:50, w * 70:w * 70 + 50] = 1.0
    return img


EXPECTED_WORD_BOXES_TLHW = [  # lib.rs:437-445
    (27.0, -3.0, 25.0, 56.0),
    (27.0, 66.0, 25.0, 57.0),
    (27.0, 136.0, 25.0, 57.0),
]

# lib.rs:34 with the EUR sign restored (see lib.rs:33)
DEFAULT_ALPHABET = " 0123456789!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~€ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"


def make_alphabet():
    """lib.rs:424-427."""
    return DEFAULT_ALPHABET[:63]


def fake_detection_run(x):
    """FakeDetectionModel::run, lib.rs:355-361."""
    return (x + np.float32(0.5)).astype(np.float32)


FAKE_DETECTION_SHAPE = [None, 1, 200, 100]  # lib.rs:343-353


def fake_recognition_run(nchw):
    """FakeRecognitionModel::run, lib.rs:387-421: max-pool width by 4, -> [W/4, N, 64]."""
    n, c, h, w = nchw.shape
    assert c == 1 and h == 64
    wb = w // 4
    out = nchw[:, 0, :, :wb * 4].reshape(n, h, wb, 4).max(axis=3)  # [n,h,wb]
    return np.ascontiguousarray(out.transpose(2, 0, 1)).astype(np.float32)


FAKE_RECOGNITION_SHAPE = [None, 1, 64, None]  # lib.rs:376-385


def xorshift_shuffle(items, seed=1234):
    """Deterministic shuffle standing in for fastrand::Rng::with_seed(1234).shuffle
    (layout_analysis.rs:328-329): the KAT only needs *a* fixed permutation."""
    rng = np.random.default_rng(seed)
    idx = rng.permutation(len(items))
    return [items[i] for i in idx]
