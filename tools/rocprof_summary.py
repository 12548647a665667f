#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (rocprofv3 --kernel-trace --stats -o X)
into a per-kernel table: calls, total/avg/min/max duration, share.  Used to produce the
summaries committed under profiles/."""
import re
import sqlite3
import sys


def main(path, out=None, skip_first=0):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = db.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, kd, ks)).fetchall()
    stats = {}
    for name, st, en in rows:
        name = re.sub(r"\(.*", "", name)
        name = re.sub(r"^void ", "", name)
        d = (en - st) / 1e3  # us
        s = stats.setdefault(name, [0, 0.0, 1e30, 0.0])
        s[0] += 1
        s[1] += d
        s[2] = min(s[2], d)
        s[3] = max(s[3], d)
    total = sum(s[1] for s in stats.values())
    lines = ["# rocprofv3 --kernel-trace --stats summary of %s" % path,
             "# columns: kernel, calls, total_us, avg_us, min_us, max_us, percent",
             "%-64s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%")]
    for name, s in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        lines.append("%-64s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%" % (name[:64], s[0], s[1], s[1] / s[0], s[2], s[3], 100 * s[1] / total))
    lines.append("TOTAL kernel time %.1f us over %d dispatches" % (total, len(rows)))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
