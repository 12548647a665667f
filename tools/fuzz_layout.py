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
from oracle.geometry import RotatedRect
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
    rng=np.random.default_rng(1000+seed)
    words=[]
    mode=seed%4
    cols=int(rng.integers(1,4))
    for c in range(cols):
        x0=20+c*int(rng.integers(250,400)); y=20
        for _ in range(int(rng.integers(5,30))):
            h=int(rng.integers(8,40 if mode==1 else 22))
            x=x0+int(rng.integers(0,30))
            for _ in range(int(rng.integers(1,10))):
                w=int(rng.integers(4,200 if mode==2 else 70))
                if x+w>x0+ (600 if mode==2 else 300): break
                ang=float(rng.normal(0,0.25 if mode==3 else 0.03))
                up=(np.float32(np.sin(ang)),np.float32(np.cos(ang)))
                words.append(RotatedRect.new((np.float32(x+w/2+rng.uniform(-2,2)),np.float32(y+h/2+rng.uniform(-3,3))),up,np.float32(w+rng.uniform(0,8)),np.float32(h+rng.uniform(0,8))))
                x+=w+int(rng.integers(-3,20))
            y+=h+int(rng.integers(-2,30))
    words=[words[i] for i in rng.permutation(len(words))]
    got=host(words); exp=oracle_ftl(words)
    ok=len(got)==len(exp) and all(np.array_equal(g,np.array([w.to_array() for w in e],np.float32).reshape(-1,6)) for g,e in zip(got,exp))
    if not ok: bad+=1; print("MISMATCH seed",seed,len(words))
print("seeds",sys.argv[1],sys.argv[2],"bad",bad,"%.0fs"%(time.time()-t0))
