"""Oracle restatement of ocrs/src/layout_analysis.rs and
ocrs/src/layout_analysis/empty_rects.rs (words -> lines in reading order).

TEST INFRASTRUCTURE ONLY.  Pinned by the reference KATs
layout_analysis.rs:243-350 and empty_rects.rs:239-294 (tests/test_oracle_kat.py).
"""
import math

import numpy as np

from .geometry import (Line, Rect, as_i32, bounding_rect_of, f32, leftmost_edge, rightmost_edge, rround)


class RustBinaryHeap:
    """std::collections::BinaryHeap (max-heap) with the exact sift order of the
    Rust standard library, so that ties between equal scores pop in the same
    order as in empty_rects.rs:41,91,128.  Items are (score, payload); only
    `score` is compared (empty_rects.rs:20-24, f32::total_cmp)."""

    def __init__(self):
        self.data = []

    @staticmethod
    def _le(a, b):
        return a[0] <= b[0]

    def push(self, item):
        self.data.append(item)
        self._sift_up(0, len(self.data) - 1)

    def _sift_up(self, start, pos):
        d = self.data
        elt = d[pos]
        while pos > start:
            parent = (pos - 1) // 2
            if self._le(elt, d[parent]):
                break
            d[pos] = d[parent]
            pos = parent
        d[pos] = elt
        return pos

    def pop(self):
        d = self.data
        if not d:
            return None
        item = d.pop()
        if d:
            item, d[0] = d[0], item
            self._sift_down_to_bottom(0)
        return item

    def _sift_down_to_bottom(self, pos):
        d = self.data
        end = len(d)
        start = pos
        elt = d[pos]
        child = 2 * pos + 1
        while child <= max(end - 2, 0) and child + 1 < end:
            if self._le(d[child], d[child + 1]):
                child += 1
            d[pos] = d[child]
            pos = child
            child = 2 * pos + 1
        if child == end - 1:
            d[pos] = d[child]
            pos = child
        d[pos] = elt
        self._sift_up(start, pos)


def max_empty_rects(obstacles, boundary, score, min_width, min_height):
    """empty_rects.rs:47-138 — generator of maximal empty rects by descending score."""
    obs = sorted(obstacles, key=lambda o: o.center())  # (c.x, c.y), stable
    heap = RustBinaryHeap()
    if not boundary.is_empty():
        heap.push((score(boundary), boundary, obs))
    while True:
        part = heap.pop()
        if part is None:
            return
        _, b, obstacles_p = part
        if not obstacles_p:
            yield b
            continue
        pivot = obstacles_p[len(obstacles_p) // 2]
        right_rect = Rect.from_tlbr(b.top, pivot.right, b.bottom, b.right)
        left_rect = Rect.from_tlbr(b.top, b.left, b.bottom, pivot.left)
        top_rect = Rect.from_tlbr(b.top, b.left, pivot.top, b.right)
        bottom_rect = Rect.from_tlbr(pivot.bottom, b.left, b.bottom, b.right)
        for sr in (top_rect, left_rect, bottom_rect, right_rect):
            if max(sr.width(), 0) < min_width or max(sr.height(), 0) < min_height or sr.is_empty():
                continue
            sr_obs = [o for o in obstacles_p if o.intersects(sr)]
            assert len(sr_obs) < len(obstacles_p)
            heap.push((score(sr), sr, sr_obs))


def filter_overlapping(source, factor):
    """empty_rects.rs:184-221."""
    found = []
    for r in source:
        if any(f.iou(r) >= factor for f in found):
            continue
        found.append(r)
        yield r


class _W:
    """Per-word cached quantities for group_into_lines."""

    __slots__ = ("rect", "cx", "left_i", "ledge", "redge", "ledge_cx", "redge_cx", "cx_i")

    def __init__(self, r):
        self.rect = r
        self.cx = r.cx
        self.left_i = as_i32(r.bounding_rect().left)
        self.ledge = leftmost_edge(r)
        self.redge = rightmost_edge(r)
        self.ledge_cx = self.ledge.center()[0]
        self.redge_cx = self.redge.center()[0]
        self.cx_i = as_i32(r.cx)


def group_into_lines(rects, separators):
    """layout_analysis.rs:19-71."""
    ws = sorted((_W(r) for r in rects), key=lambda w: w.left_i)  # stable
    lines = []
    overlap_threshold = f32(5)
    max_h_overlap = f32(5)
    while ws:
        line = [ws.pop(0)]
        while True:
            last = line[-1]
            best_i = -1
            best_key = None
            for i, w in enumerate(ws):
                if not (w.cx > last.cx):
                    continue
                if not (w.ledge_cx - last.redge_cx >= -max_h_overlap):
                    continue
                if not (last.redge.vertical_overlap(w.ledge) >= overlap_threshold):
                    continue
                if separators:
                    a_to_b = Line(last.rect.center(), w.rect.center())
                    if any(a_to_b.intersects(s) for s in separators):
                        continue
                if best_key is None or w.cx_i < best_key:  # min_by_key: first minimum
                    best_key = w.cx_i
                    best_i = i
            if best_i < 0:
                break
            line.append(ws.pop(best_i))
        lines.append([w.rect for w in line])
    return lines


def find_block_separators(words):
    """layout_analysis.rs:83-155."""
    br = bounding_rect_of(w.bounding_rect() for w in words)
    if br is None:
        return []
    page_rect = br.integral_bounding_rect()

    lines = group_into_lines(words, [])
    lines.sort(key=lambda l: int(rround(l[0].bounding_rect().top)))

    all_spacings = []
    for line in lines:
        if len(line) > 1:
            spacings = []
            for cur, nxt in zip(line, line[1:]):
                v = nxt.bounding_rect().left - cur.bounding_rect().right
                v = v if v > 0 else f32(0.0)  # f32::max(0.)
                spacings.append(int(rround(v)))
            spacings.sort()
            all_spacings.extend(spacings)
    all_spacings.sort()
    median_word_spacing = all_spacings[len(all_spacings) // 2] if all_spacings else 10
    median_height = int(rround(words[len(words) // 2].height())) if words else 10

    def score(r):
        aspect = f32(r.height()) / f32(r.width())
        lg = abs(f32(math.log2(float(aspect)))) if aspect > 0 else f32(np.inf)
        if lg < 3.0:
            wgt = f32(0.5)
        elif lg < 5.0:
            wgt = f32(1.5)
        else:
            wgt = f32(lg)
        return f32(np.sqrt(f32(r.area()) * wgt))

    object_bboxes = [w.bounding_rect().integral_bounding_rect() for w in words]
    min_width = median_word_spacing * 3
    min_height = 3 * max(median_height, 0)
    assert min_width >= 0  # `.try_into().unwrap()` on u32

    out = []
    for r in filter_overlapping(max_empty_rects(object_bboxes, page_rect, score, min_width, min_height), f32(0.5)):
        out.append(r)
        if len(out) >= 80:
            break
    return out


def find_text_lines(words):
    """layout_analysis.rs:158-233.  words: list[RotatedRect] -> list[list[RotatedRect]]."""
    separators = find_block_separators(words)
    vertical, horizontal = [], []
    for r in separators:
        cx, cy = r.center()
        vertical.append(Line((f32(cx), f32(r.top)), (f32(cx), f32(r.bottom))))
        horizontal.append(Line((f32(r.left), f32(cy)), (f32(r.right), f32(cy))))

    lines = group_into_lines(words, vertical)

    def midpoint_line(ws):
        assert ws
        return Line(ws[0].bounding_rect().left_edge().center(), ws[-1].bounding_rect().right_edge().center())

    lines.sort(key=lambda ws: as_i32(midpoint_line(ws).center()[1]))

    def is_separated_by(a, b, seps):
        a_to_b = Line(a.center(), b.center())
        return any(s.intersects(a_to_b) for s in seps)

    paragraphs = []
    while lines:
        seed = lines.pop(0)
        para = [seed]
        prev = midpoint_line(seed)
        idx = 0
        while idx < len(lines):
            cand = midpoint_line(lines[idx])
            if prev.horizontal_overlap(cand) > 0 and not is_separated_by(prev, cand, horizontal):
                para.append(lines.pop(idx))
                prev = cand
            else:
                idx += 1
        paragraphs.append(para)
    return [line for para in paragraphs for line in para]
