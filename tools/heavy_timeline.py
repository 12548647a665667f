#!/usr/bin/env python
"""Timeline of the conv-stack stream in a rocprofv3 --kernel-trace of the pipelined bench.

    python tools/heavy_timeline.py <results.db> [first_step last_step]

Per step (conv1 launch to the next conv1 launch): period, busy time on the stream, idle time and where the idle
time sits (before which kernel), plus the kernels of OTHER streams that overlap the idle gaps.
"""
import sqlite3
import sys
from collections import defaultdict

from trace_occupancy import family


def main(path, lo=None, hi=None):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = db.execute("select s.%s, d.stream_id, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start"
                      % (name_col, kd, ks)).fetchall()
    t0 = rows[0][2]
    conv_stream = defaultdict(float)
    for n, st, s, e in rows:
        if "conv3x3_ragged" in n:
            conv_stream[st] += e - s
    cs = max(conv_stream.items(), key=lambda kv: kv[1])[0]
    heavy = [(s, e, family(n)) for n, st, s, e in rows if st == cs]
    starts = [i for i, (s, e, f) in enumerate(heavy) if f in ("conv1_relu_pool", "conv12_fused")]   # first kernel of a request's conv stack
    lo = 0 if lo is None else lo
    hi = len(starts) - 1 if hi is None else min(hi, len(starts) - 1)
    print("conv-stack stream %d: %d steps in the trace; steps %d..%d" % (cs, len(starts), lo, hi))
    tot_period = tot_busy = 0.0
    gap_before = defaultdict(float)
    fam_busy = defaultdict(float)
    nsteps = 0
    for k in range(lo, hi):
        a, b = starts[k], starts[k + 1]
        period = (heavy[b][0] - heavy[a][0]) / 1e6
        busy = sum(e - s for s, e, f in heavy[a:b]) / 1e6
        for s, e, f in heavy[a:b]:
            fam_busy[f] += (e - s) / 1e6
        for i in range(a, b):
            g = (heavy[i + 1][0] - heavy[i][1]) / 1e6
            if g > 0:
                gap_before[heavy[i + 1][2] + " after " + heavy[i][2]] += g
        tot_period += period
        tot_busy += busy
        nsteps += 1
        print("  step %3d at %8.1f ms: period %6.2f  busy %6.2f  idle %6.2f" % (k, (heavy[a][0] - t0) / 1e6, period, busy, period - busy))
    if nsteps:
        print("mean period %.2f ms, busy %.2f, idle %.2f" % (tot_period / nsteps, tot_busy / nsteps, (tot_period - tot_busy) / nsteps))
        print("busy per step by family:", ", ".join("%s %.2f" % (f, v / nsteps) for f, v in sorted(fam_busy.items(), key=lambda kv: -kv[1])))
        print("idle per step by position:")
        for kpos, v in sorted(gap_before.items(), key=lambda kv: -kv[1])[:8]:
            print("   %-60s %.2f ms" % (kpos, v / nsteps))


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:4]))
