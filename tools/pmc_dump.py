#!/usr/bin/env python
"""Dump every counter of rocprofv3 --pmc result databases per kernel (per-launch averages).
Usage: tools/pmc_dump.py <results.db>... [--match substr]"""
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from pmc_summary import load

match = None
paths = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--match":
        match = args.pop(0)
    else:
        paths.append(a)
for p in paths:
    out, launches, dur = load(p)
    print("#", p)
    for name in sorted(out, key=lambda n: -dur[n]):
        if match and match not in name:
            continue
        n = launches[name]
        print("%s  launches=%d avg_us=%.1f" % (name[:70], n, dur[name] / n / 1e3))
        for c, v in sorted(out[name].items()):
            print("    %-32s %16.0f /launch" % (c, v / n))
