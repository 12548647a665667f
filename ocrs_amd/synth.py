"""Synthetic workloads named by BASELINE.json / SURVEY.md §8(d) (no network, so no
real scans or weights): seeded pages of dark word blobs on a light background,
and seeded 64x256 text-line crops."""
import numpy as np


def synthetic_page(seed, height=1024, width=1024, lines=80, columns=2, channels=3):
    """u8 [H,W,C] page: white background, ~`lines` text lines of height 12-18 px in
    `columns` columns, 6-10 dark rounded "words" per line, Gaussian noise sigma=4."""
    rng = np.random.default_rng(seed)
    img = np.full((height, width), 255.0, np.float32)
    rows = max(1, lines // columns)
    margin = 16
    col_w = (width - margin * (columns + 1)) // columns
    pitch = (height - 2 * margin) / rows
    for c in range(columns):
        x_left = margin + c * (col_w + margin)
        for r in range(rows):
            lh = int(rng.integers(12, max(13, min(19, int(pitch) - 6))))
            y0 = int(margin + r * pitch + rng.integers(0, max(1, int(pitch) - lh - 5)))
            n_words = int(rng.integers(6, 11))
            x = x_left + int(rng.integers(0, 12))
            for _ in range(n_words):
                ww = int(rng.integers(22, 58))
                if x + ww >= x_left + col_w:
                    break
                shade = float(rng.integers(0, 70))
                wh = lh - int(rng.integers(0, 3))
                yy = y0 + int(rng.integers(0, 2))
                img[yy:yy + wh, x:x + ww] = shade
                # round the corners
                for (cy, cx) in ((yy, x), (yy, x + ww - 1), (yy + wh - 1, x), (yy + wh - 1, x + ww - 1)):
                    img[cy, cx] = 255.0
                # glyph-like light slits so the crops are not flat
                for sx in range(x + 3, x + ww - 3, int(rng.integers(4, 8))):
                    img[yy + 2:yy + wh - 2, sx] = shade + 90.0
                x += ww + int(rng.integers(9, 15))
    img += rng.normal(0.0, 4.0, img.shape).astype(np.float32)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if channels == 1:
        return img[:, :, None].copy()
    out = np.repeat(img[:, :, None], channels, axis=2)
    if channels == 4:
        out[:, :, 3] = 255
    return np.ascontiguousarray(out)


def synthetic_line_crops(seed, n=2048, height=64, width=256):
    """f32 [n, height, width] in [-0.5, 0.5]: seeded glyph-like strokes (config 3)."""
    rng = np.random.default_rng(seed)
    out = np.full((n, height, width), 0.5, np.float32)
    for i in range(n):
        x = int(rng.integers(2, 10))
        while x < width - 12:
            gw = int(rng.integers(5, 14))
            top = int(rng.integers(8, 24))
            bot = int(rng.integers(40, 58))
            out[i, top:bot, x:x + max(2, gw // 3)] = -0.5
            if rng.random() < 0.6:
                mid = int(rng.integers(top + 4, bot - 4))
                out[i, mid:mid + 3, x:x + gw] = -0.45
            if rng.random() < 0.4:
                out[i, top:top + 3, x:x + gw] = -0.5
            x += gw + int(rng.integers(3, 9))
            if rng.random() < 0.15:
                x += int(rng.integers(8, 16))
    out += rng.normal(0.0, 0.015, out.shape).astype(np.float32)
    return np.clip(out, -0.5, 0.5).astype(np.float32)
