#!/usr/bin/env python
"""Per-kernel HBM traffic / MFMA-busy summary from the rocprofv3 --pmc passes of tools/profile.sh.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches — FETCH_SIZE on gfx950 reports half of
the bytes of a wide (16 B/lane) coalesced read stream (MI355X_MICROARCH.md §HBM), WRITE_SIZE is taken as is;
both are in KiB and are collected in separate passes (they do not fit one pass)."""
import collections
import json
import re
import sqlite3
import sys


def load(path):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tables if t.startswith(p)][0]
    kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    dur = collections.Counter()
    for name, val, cname, d, eid in db.execute(
            "select s.%s, p.value, i.name, d.end-d.start, d.id from %s p join %s d on p.event_id = d.event_id "
            "join %s s on d.kernel_id=s.id join %s i on p.pmc_id = i.id" % (name_col, pe, kd, ks, pi)):
        name = re.sub(r"\(.*", "", name)
        out[name][cname] += val
    for name, d in db.execute("select s.%s, d.end-d.start from %s d join %s s on d.kernel_id=s.id" % (name_col, kd, ks)):
        name = re.sub(r"\(.*", "", name)
        launches[name] += 1
        dur[name] += d
    return out, launches, dur


def main(prefix, out_txt, out_json, requests=None):
    fetch, lf, df = load(prefix + "/fetch_results.db")
    write, lw, dw = load(prefix + "/write_results.db")
    mfma, lm, dm = load(prefix + "/mfma_results.db")
    rows = []
    for name in fetch:
        n = lf[name]
        f_kib = fetch[name].get("FETCH_SIZE", 0.0)
        w_kib = write.get(name, {}).get("WRITE_SIZE", 0.0)
        nw = max(lw.get(name, n), 1)
        hbm = (2.0 * f_kib / n + w_kib / nw) * 1024.0
        mb = mfma.get(name, {})
        busy = mb.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        avg_us = dm.get(name, 0) / max(lm.get(name, 1), 1) / 1e3
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs (256 CUs x 4)
        gui = mb.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        mfma_util = busy / (gui * 1024.0) if gui > 0 else None
        clock_ghz = gui / max(dm.get(name, 0), 1) if gui > 0 else None  # cycles per ns
        rows.append(dict(kernel=name, launches=n, fetch_kib_per_launch=f_kib / n, write_kib_per_launch=w_kib / nw,
                         hbm_bytes_per_launch=hbm, avg_us_in_pmc_pass=avg_us,
                         mfma_busy_cycles_per_launch=busy / max(lm.get(name, 1), 1),
                         mfma_insts_per_launch=mb.get("SQ_INSTS_MFMA", 0.0) / max(lm.get(name, 1), 1),
                         mfma_pipe_util=mfma_util, shader_clock_ghz=clock_ghz))
    rows.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
    lines = ["# HBM traffic per launch from rocprofv3 --pmc passes (FETCH_SIZE x2 correction for gfx950, see header of tools/pmc_summary.py)",
             "# mfma% = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); GHz = GRBM_GUI_ACTIVE/8 / duration",
             "%-62s %8s %14s %14s %14s %10s %7s %6s" % ("kernel", "launches", "fetch KiB/l", "write KiB/l", "HBM MB/launch", "avg us", "mfma%", "GHz")]
    for r in rows:
        lines.append("%-62s %8d %14.1f %14.1f %14.2f %10.1f %7s %6s" % (
            r["kernel"][:62], r["launches"], r["fetch_kib_per_launch"], r["write_kib_per_launch"],
            r["hbm_bytes_per_launch"] / 1e6, r["avg_us_in_pmc_pass"],
            "%.1f" % (100 * r["mfma_pipe_util"]) if r["mfma_pipe_util"] is not None else "-",
            "%.2f" % r["shader_clock_ghz"] if r["shader_clock_ghz"] is not None else "-"))
    open(out_txt, "w").write("\n".join(lines) + "\n")
    table = {r["kernel"]: r for r in rows}
    if requests:   # how many requests the profiled loop ran (bench.py detection_traffic divides the summed bytes by it)
        table["_meta"] = {"requests": int(requests)}
    json.dump(table, open(out_json, "w"), indent=1)
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
