"""Random pages of word rects for the layout analysis (shared by tools/fuzz_layout.py and the tests): 1-3 columns of
small / tall / wide / rotated words, shuffled.  fuzz_page(seed) -> list of oracle.geometry.RotatedRect."""
import numpy as np

from oracle.geometry import RotatedRect


def fuzz_page(seed):
    rng = np.random.default_rng(1000 + seed)
    words = []
    mode = seed % 4
    cols = int(rng.integers(1, 4))
    for c in range(cols):
        x0 = 20 + c * int(rng.integers(250, 400))
        y = 20
        for _ in range(int(rng.integers(5, 30))):
            h = int(rng.integers(8, 40 if mode == 1 else 22))
            x = x0 + int(rng.integers(0, 30))
            for _ in range(int(rng.integers(1, 10))):
                w = int(rng.integers(4, 200 if mode == 2 else 70))
                if x + w > x0 + (600 if mode == 2 else 300):
                    break
                ang = float(rng.normal(0, 0.25 if mode == 3 else 0.03))
                up = (np.float32(np.sin(ang)), np.float32(np.cos(ang)))
                words.append(RotatedRect.new((np.float32(x + w / 2 + rng.uniform(-2, 2)), np.float32(y + h / 2 + rng.uniform(-3, 3))),
                                             up, np.float32(w + rng.uniform(0, 8)), np.float32(h + rng.uniform(0, 8))))
                x += w + int(rng.integers(-3, 20))
            y += h + int(rng.integers(-2, 30))
    return [words[i] for i in rng.permutation(len(words))]


def words_array(words):
    return np.ascontiguousarray(np.array([w.to_array() for w in words], np.float32).reshape(-1, 6))
