#!/usr/bin/env python
"""Where does the GPU's time go in a rocprofv3 --kernel-trace of the pipelined bench?

    python tools/trace_occupancy.py <results.db> [t_skip_fraction]

Prints, for the steady-state part of the trace (the first `t_skip_fraction`, default 0.45, is warm-up):
  * wall span, the union of all kernel intervals (GPU busy), and per stream: busy time and kernel count;
  * per kernel family: summed duration, and the share of wall time during which at least one such kernel runs;
  * the idle gaps on the stream that carries the conv stacks (the longest ones, with what ran elsewhere meanwhile).
"""
import re
import sqlite3
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def family(name):
    name = re.sub(r"\(.*", "", name)
    for key in ("conv12_fused", "conv3x3_ragged", "gru_persistent", "gru_step_fused", "gemm_tiled", "gemm_mfma", "double_conv", "dwpw_fused",
                "dwconv3x3", "conv1_relu_pool", "crop_lines", "contour_rect", "trace_count", "ccl_", "log_softmax",
                "ctc_collapse", "copyBuffer", "fillBuffer", "pool", "resize", "prepare_image"):
        if key in name:
            return key
    return name[:40]


def main(path, skip=0.45):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = db.execute("select s.%s, d.stream_id, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start"
                      % (name_col, kd, ks)).fetchall()
    t0, t1 = rows[0][2], max(r[3] for r in rows)
    cut = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[2] >= cut]
    t0, t1 = rows[0][2], max(r[3] for r in rows)
    wall = (t1 - t0) / 1e6
    print("steady-state window: %.1f ms, %d dispatches" % (wall, len(rows)))
    print("GPU busy (union of all kernels): %.1f ms = %.1f %%" % (union([(r[2], r[3]) for r in rows]) / 1e6,
                                                                  100 * union([(r[2], r[3]) for r in rows]) / 1e6 / wall))
    by_stream = defaultdict(list)
    fam = defaultdict(list)
    for n, st, s, e in rows:
        by_stream[st].append((s, e, n))
        fam[family(n)].append((s, e))
    print("\nper stream: busy ms (share of wall), kernels, dominant family")
    conv_stream = None
    for st, iv in sorted(by_stream.items(), key=lambda kv: -union([(a, b) for a, b, _ in kv[1]])):
        busy = union([(a, b) for a, b, _ in iv]) / 1e6
        fams = defaultdict(float)
        for a, b, n in iv:
            fams[family(n)] += (b - a) / 1e6
        dom = max(fams.items(), key=lambda kv: kv[1])
        if dom[0] in ("conv3x3_ragged", "conv12_fused") and conv_stream is None:
            conv_stream = st
        print("  stream %3d: %8.1f ms (%5.1f %%)  %6d kernels  %s %.1f ms" % (st, busy, 100 * busy / wall, len(iv), dom[0], dom[1]))
    print("\nper kernel family: summed ms, share of wall with >= 1 such kernel running, launches")
    for k, iv in sorted(fam.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
        print("  %-22s %9.1f ms  %5.1f %%  %6d" % (k, sum(b - a for a, b in iv) / 1e6, 100 * union(iv) / 1e6 / wall, len(iv)))
    if conv_stream is not None:
        iv = sorted((a, b, n) for a, b, n in by_stream[conv_stream])
        gaps = []
        for (a0, b0, n0), (a1, b1, n1) in zip(iv, iv[1:]):
            if a1 > b0:
                gaps.append((a1 - b0, b0, a1, family(n0), family(n1)))
        tot_gap = sum(g[0] for g in gaps) / 1e6
        print("\nconv-stack stream %d: %.1f ms idle in %d gaps (%.1f %% of wall); largest:" % (conv_stream, tot_gap, len(gaps), 100 * tot_gap / wall))
        for g in sorted(gaps, reverse=True)[:8]:
            others = defaultdict(float)
            for n, st, s, e in rows:
                if st != conv_stream and e > g[1] and s < g[2]:
                    others[family(n)] += (min(e, g[2]) - max(s, g[1])) / 1e6
            top = ", ".join("%s %.2f" % kv for kv in sorted(others.items(), key=lambda kv: -kv[1])[:3])
            print("   %.2f ms after %s before %s; meanwhile: %s" % (g[0] / 1e6, g[3], g[4], top))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.45)
