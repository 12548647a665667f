#!/usr/bin/env python
"""Detection chains (resize_pages .. contour_rect on one stream) in a kernel trace of the pipelined bench: how long a
request's detection stage takes beside the conv stacks of other requests, and the stalls of the conv-stack stream.

    python tools/det_chains.py <results.db>
"""
import sqlite3
import sys
from collections import defaultdict

from trace_occupancy import family


def main(path):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = db.execute("select s.%s, d.stream_id, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start"
                      % (name_col, kd, ks)).fetchall()
    t0 = rows[0][2]
    conv = defaultdict(float)
    for n, st, s, e in rows:
        if "conv3x3_ragged" in n:
            conv[st] += e - s
    cs = max(conv.items(), key=lambda kv: kv[1])[0]
    heavy = [(s, e) for n, st, s, e in rows if st == cs]
    stalls = [(heavy[i][1], heavy[i + 1][0]) for i in range(len(heavy) - 1) if heavy[i + 1][0] - heavy[i][1] > 5e6]
    print("conv-stack stream %d stalls > 5 ms: %s" % (cs, ", ".join("%.0f-%.0f" % ((a - t0) / 1e6, (b - t0) / 1e6) for a, b in stalls)))
    cur, chains = {}, []
    slow = defaultdict(lambda: [0, 0.0])
    for n, st, s, e in rows:
        if st == cs:
            continue
        if "resize_pages" in n:
            cur[st] = [s, 0, []]
        if st in cur:
            cur[st][1] += 1
            cur[st][2].append((family(n), (e - s) / 1e6))
            if "contour_rect" in n:
                chains.append(((cur[st][0] - t0) / 1e6, (e - t0) / 1e6, st, cur[st][1], cur[st][2]))
                del cur[st]
    durs = []
    for a, b, st, k, ks_ in chains:
        durs.append(b - a)
        for f, d in ks_:
            slow[f][0] += 1
            slow[f][1] += d
    durs.sort()
    if durs:
        print("%d detection chains: median %.1f ms, mean %.1f, p90 %.1f, max %.1f" % (len(durs), durs[len(durs) // 2], sum(durs) / len(durs),
                                                                                    durs[int(len(durs) * 0.9)], durs[-1]))
        print("mean time per kernel family inside a chain (dispatch to end), ms per chain:")
        for f, (c, d) in sorted(slow.items(), key=lambda kv: -kv[1][1])[:12]:
            print("   %-28s %7.2f   (%d launches per chain)" % (f[:28], d / len(chains), round(c / len(chains))))


if __name__ == "__main__":
    main(sys.argv[1])
