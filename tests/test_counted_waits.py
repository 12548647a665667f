"""The counted `s_waitcnt vmcnt(N)` of the bf16-split kernels against the instructions the compiler really emitted.

split::pipeline (split_mfma.hpp: conv3x3_ragged_kernel<..., NP>, gemm_split_kernel<NP>) and conv12_fused_split_kernel<NP>
release their barriers behind "at most N vector-memory instructions outstanding" — correct only while load_a stays four
instructions, load_b NP LDS-DMA instructions, in program order.  tools/counted_waits.py disassembles the gfx950 code
objects of the build that ships, walks every kernel's control-flow graph and checks every path into every counted wait
(DESIGN.md §6.5).  No GPU needed: hipcc cross-compiles.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import counted_waits as CW  # noqa: E402


@pytest.fixture(scope="module")
def rows_errors():
    from ocrs_amd import build
    build.build()
    if not os.path.exists(CW.OBJDUMP):
        pytest.skip("llvm-objdump not found at %s" % CW.OBJDUMP)
    return CW.check_all()


def test_no_counted_wait_leaves_the_copy_it_guards_in_flight(rows_errors):
    rows, errors = rows_errors
    assert not errors, "\n".join(errors)


def test_every_split_kernel_still_has_its_counted_waits(rows_errors):
    """A build in which the waits no longer compile to `s_waitcnt vmcnt(N)` + `s_barrier` would pass the test above vacuously."""
    rows, _ = rows_errors
    per_kernel = {}
    for name, addr, n, np_, hists, slack in rows:
        per_kernel.setdefault(name, []).append((n, np_, slack))
    conv = [k for k in per_kernel if "conv3x3_ragged_kernel" in k]
    gemm = [k for k in per_kernel if "gemm_split_kernel" in k]
    c12 = [k for k in per_kernel if "conv12_fused_split_kernel" in k]
    assert len(conv) == 24 and len(gemm) == 2 and len(c12) == 2, (len(conv), len(gemm), len(c12))
    for k in conv + gemm:
        # the steady-state body ends its four half-steps with vmcnt(2 NP + 4), (2 NP + 8), (2 NP + 4), (2 NP + 8); the compiler
        # may stand a stricter wait of its own in front of a barrier (for activation registers the next commit reads): then
        # that one is listed, with its slack
        np_ = per_kernel[k][0][1]
        assert len(per_kernel[k]) == 4, (k, per_kernel[k])
        assert all(n <= 2 * np_ + 8 and (s is None or s >= 0) for n, _, s in per_kernel[k]), (k, per_kernel[k])
        # and the pipeline really is counted: with the MUBUF-form LDS-DMA the compiler no longer stands a vmcnt(0) in front of
        # the waits (round 5's FLAT-form build: 105 of 118 behind such a drain) — at least two of the four
        # guards are exactly as the source counts them, none behind a drain
        assert sum(1 for n, _, s in per_kernel[k] if s == 0) >= 2 and all(s is not None for _, _, s in per_kernel[k]), (k, per_kernel[k])
    for k in c12:              # seven taps end with (at most) vmcnt(NP), the last two drain
        np_ = per_kernel[k][0][1]
        assert len(per_kernel[k]) == 7 and all(n <= np_ for n, _, _ in per_kernel[k]), (k, per_kernel[k])


def test_the_checker_catches_a_short_wait():
    """The analysis itself, on hand-made instruction streams: a load_b of two instead of three instructions, and a load
    moved across the wait's reach, must be flagged; spills (more instructions) must not."""
    def stream(body):
        ins, a = [], 0x1000
        for mn, ops in body:
            ins.append((a, mn, ops))
            a += 4
        return ins
    L, A = ("global_load_lds_dwordx4", "v[0:1], off"), ("buffer_load_dwordx4", "v[4:7], v1, s[0:3], 0 offen")
    S = ("scratch_store_dword", "off, v9, off")
    wait = lambda n: [("s_waitcnt", "vmcnt(%d)" % n), ("s_waitcnt", "lgkmcnt(0)"), ("s_barrier", "")]
    B = ("s_barrier", "")
    good = [L] * 3 + [B] + [L] * 3 + [A] * 4 + [B] + [L] * 3 + wait(10) + [("s_endpgm", "")]
    (_, n, hists), = CW.analyse(stream(good), depth=29)
    assert n == 10 and [CW.younger_than_needed(h, 7) for h in hists] == [10]
    short = [L] * 3 + [B] + [L] * 2 + [A] * 4 + [B] + [L] * 3 + wait(10) + [("s_endpgm", "")]          # a load_b lost an instruction
    (_, n, hists), = CW.analyse(stream(short), depth=29)
    assert any(CW.short_weight_copy(h, 3) for h in hists)    # (counting LDS-DMA instructions back would name the wrong one)
    assert not any(CW.short_weight_copy(h, 3) for h in CW.analyse(stream(good), depth=29)[0][2])
    moved = [L] * 3 + [B] + [L] * 2 + [A] * 4 + [B] + [L] * 4 + wait(10) + [("s_endpgm", "")]          # ... or had one moved across a barrier
    assert any(CW.short_weight_copy(h, 3) for h in CW.analyse(stream(moved), depth=29)[0][2])
    (_, n, hists), = CW.analyse(stream([L] * 3 + [L] * 3 + [A] * 2 + [L] * 3 + wait(10) + [("s_endpgm", "")]), depth=29)
    assert min(CW.younger_than_needed(h, 7) for h in hists) == 8 < 10
    spilled = [L] * 3 + [L] * 3 + [A] * 4 + [S] * 3 + [L] * 3 + wait(10) + [("s_endpgm", "")]
    (_, n, hists), = CW.analyse(stream(spilled), depth=29)
    assert [CW.younger_than_needed(h, 7) for h in hists] == [13]
    # a conditional load before the wait: both paths are examined
    branchy = [L] * 3 + [L] * 3 + [("s_cbranch_scc1", "4")] + [A] * 4 + [L] * 3 + wait(10) + [("s_endpgm", "")]
    (_, n, hists), = CW.analyse(stream(branchy), depth=29)
    assert sorted(CW.younger_than_needed(h, 7) for h in hists) == [6, 10]
