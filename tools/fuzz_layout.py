#!/usr/bin/env python
"""Randomised check of the host layout analysis (layout.cpp) against the oracle (oracle/layout.py): pages of 1-3 columns
with small / tall / wide / rotated words, shuffled; any difference in the lines (order, membership, bits) is reported.
CPU only.    python tools/fuzz_layout.py <first seed> <last seed (exclusive)>"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ocrs_amd import _lib
from oracle.layout import find_text_lines as oracle_ftl
from oracle.geometry import RotatedRect  # noqa: F401
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from fuzz_pages import fuzz_page  # noqa: E402
lib=_lib.lib()
def host(words):
    a = np.ascontiguousarray(np.array([w.to_array() for w in words], np.float32).reshape(-1, 6))
    lr = C.POINTER(C.c_float)(); lo = C.POINTER(C.c_size_t)(); nl = C.c_size_t(0)
    _lib.check(lib.ocrs_engine_find_text_lines(None, None, a.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(len(a)), C.byref(lr), C.byref(lo), C.byref(nl)))
    offs = [lo[i] for i in range(nl.value + 1)]
    flat = np.ctypeslib.as_array(lr, shape=(max(len(a), 1) * 6,))[: len(a) * 6].reshape(-1, 6).copy()
    lib.ocrs_buffer_free(lr); lib.ocrs_buffer_free(lo)
    return [flat[offs[i]:offs[i + 1]] for i in range(nl.value)]
bad=0; t0=time.time()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    words=fuzz_page(seed)
    got=host(words); exp=oracle_ftl(words)
    ok=len(got)==len(exp) and all(np.array_equal(g,np.array([w.to_array() for w in e],np.float32).reshape(-1,6)) for g,e in zip(got,exp))
    if not ok: bad+=1; print("MISMATCH seed",seed,len(words))
print("seeds",sys.argv[1],sys.argv[2],"bad",bad,"%.0fs"%(time.time()-t0))
