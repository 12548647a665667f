"""Checks the counted `s_waitcnt vmcnt(N)` of the bf16-split kernels against the code the compiler really emitted.

split::pipeline (ocrs_amd/csrc/split_mfma.hpp) and conv12_fused_split_kernel (kernels_rec.hip) end their half-steps with
`s_waitcnt vmcnt(N)` + a bare `s_barrier`, N > 0: "everything older than the N youngest vector-memory instructions has
landed".  That is only right if every load_a is exactly FOUR vector-memory instructions, every load_b exactly NP LDS-DMA
instructions, and the compiler kept them in program order.  Nothing in the language guarantees it, so this module reads
the gfx950 code object that ships inside ocrs_amd/_build/*.o (llvm-objdump), rebuilds each kernel's control-flow graph and
proves, for every counted wait followed by a barrier, over EVERY path that reaches it, that the copy the wait is there for
has landed when the barrier releases: counting back from the wait through the vector-memory instructions issued on that path
(A = load into registers, L = global_load_lds, S = store / spill), the youngest LDS-DMA instruction of the needed weight
chunk — the (2 NP + 1)-th youngest L for split::pipeline (chunks c+3 and c+2 or c+4 and c+3 may stay in flight), the
(NP + 1)-th youngest for conv12_fused_split (only the tap just requested may) — has at least N younger instructions, or lies
behind a full drain (vmcnt(0)).  MORE instructions than the source counts on (a spill, a split load) only make the wait
stricter and are reported as "over-waits"; FEWER, or a reordering, is an error.

tests/test_counted_waits.py runs it in the CPU suite; `python tools/counted_waits.py` prints the table.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "ocrs_amd", "_build")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^([0-9a-f]+) <(\S+)>:")


def device_disassembly(obj):
    """llvm-objdump -d of the gfx950 code object bundled in a host object file"""
    subprocess.run([OBJDUMP, "--offloading", obj], check=True, capture_output=True)
    co = None
    for f in os.listdir(os.path.dirname(obj)):
        if f.startswith(os.path.basename(obj) + ".") and f.endswith("gfx950"):
            co = os.path.join(os.path.dirname(obj), f)
    if co is None:
        raise RuntimeError("no gfx950 code object in %s" % obj)
    try:
        return subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
    finally:
        for f in os.listdir(os.path.dirname(obj)):
            if f.startswith(os.path.basename(obj) + "."):
                os.remove(os.path.join(os.path.dirname(obj), f))


def functions(dis):
    """{mangled name: [(addr, mnemonic, operands)]}"""
    out, cur = {}, None
    for line in dis.splitlines():
        m = _FUNC.match(line)
        if m:
            cur = out.setdefault(m.group(2), [])
            continue
        m = _INS.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def kind(mn):
    if mn.startswith("global_load_lds") or (mn.startswith("buffer_load") and " lds" in mn):
        return "L"
    if mn.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "A"
    if mn.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic", "flat_atomic")):
        return "S"
    return None


def vmcnt_of(ops):
    m = re.search(r"vmcnt\((\d+)\)", ops)
    return int(m.group(1)) if m else None


def analyse(ins, depth=32):
    """Forward data-flow over the CFG: the set of possible `depth`-long suffixes of the vector-memory issue history at every
    instruction ('.' = nothing yet, '|' = a full drain: s_waitcnt vmcnt(0), 'B' = s_barrier).  Returns [(addr, N, sorted set of histories)]
    for every counted wait (vmcnt(N), N > 0) that is followed by an s_barrier before any other vector-memory instruction."""
    idx = {a: i for i, (a, _, _) in enumerate(ins)}
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (a, mn, ops) in enumerate(ins):
        if mn == "s_endpgm":
            continue
        if mn == "s_branch" or mn.startswith("s_cbranch"):
            tgt = a + 4 + 4 * ((int(ops.split()[0]) + 0x8000) % 0x10000 - 0x8000)
            if tgt in idx:
                succ[i].append(idx[tgt])
            if mn == "s_branch":
                continue
        if mn in ("s_setpc_b64", "s_swappc_b64"):
            continue
        if i + 1 < n:
            succ[i].append(i + 1)
    # the counted waits, and the instructions from which one of them can still be reached: nothing else is tracked (the
    # epilogues' hundreds of exec-masked stores would multiply the histories without ever meeting a wait again)
    waits, guarded, strictest = [], {}, {}
    for i, (a, mn, ops) in enumerate(ins):
        v = vmcnt_of(ops) if mn == "s_waitcnt" else None
        if not v:
            continue
        j = i + 1
        drained = False
        while j < n and kind(ins[j][1] + " " + ins[j][2]) is None and ins[j][1] not in ("s_barrier", "s_endpgm") and not ins[j][1].startswith(("s_branch", "s_cbranch")):
            drained = drained or (ins[j][1] == "s_waitcnt" and vmcnt_of(ins[j][2]) == 0)
            j += 1
        if j < n and ins[j][1] == "s_barrier" and not drained:   # (a full drain in front of the barrier: nothing counted to check)
            # several waits may stand in front of one barrier with no vector-memory instruction between them (ours, and the
            # compiler's own for a register it needs, possibly merged into our lgkmcnt wait): the strictest of them holds
            if guarded.get(j) is None:
                guarded[j] = i
                waits.append(i)
                strictest[i] = v
            else:
                strictest[guarded[j]] = min(strictest[guarded[j]], v)
    pred = [[] for _ in range(n)]
    for i in range(n):
        for j in succ[i]:
            pred[j].append(i)
    live = set(waits)
    stack = list(waits)
    while stack:
        for p in pred[stack.pop()]:
            if p not in live:
                live.add(p)
                stack.append(p)
    state = [set() for _ in range(n)]
    state[0].add("." * depth)
    work = [0]
    while work:
        i = work.pop()
        a, mn, ops = ins[i]
        out = set()
        k = kind(mn + " " + ops)
        if mn == "s_waitcnt" and vmcnt_of(ops) == 0:
            k = "|"
        if mn == "s_barrier":
            k = "B"
        for h in state[i]:
            if k and not (k == "|" and h.endswith("|")):
                h = (h + k)[-depth:]
            out.add(h)
        for j in succ[i]:
            if j in live and not out <= state[j]:
                state[j] |= out
                work.append(j)
        if sum(len(x) for x in (state[j] for j in succ[i])) > 200000:
            raise RuntimeError("history sets explode at %x" % a)
    return [(ins[i][0], strictest[i], sorted(state[i])) for i in waits]


def younger_than_needed(history, nth_l):
    """number of vector-memory instructions younger than the nth_l-th youngest LDS-DMA instruction of `history`;
    None if that instruction lies behind a full drain (or before the start of the kernel): it has landed whatever N is"""
    seen = 0
    back = 0
    for k in reversed(history):
        if k in "|.":
            return None
        if k == "B":
            continue
        if k == "L":
            seen += 1
            if seen == nth_l:
                return back
        back += 1
    return back      # older than everything in view: at least this many younger instructions


def short_weight_copy(history, np_):
    """True if a barrier-to-barrier interval of `history` ('B' = s_barrier) holds a number of LDS-DMA instructions that is not
    a multiple of NP, or the interval that ends at the wait holds none (the prologue's holds three copies): a load_b that compiled to fewer or more
    than NP instructions, or was moved across a barrier.  (Other loads and compiler-made drains may sit anywhere between
    them; an interval cut off by the analysis depth is not judged.)"""
    h = history.lstrip(".")
    parts = h.split("B")
    if len(h) == len(history):
        parts = parts[1:]                  # the oldest interval is cut off
    if not parts:
        return False
    return any(p.count("L") % np_ for p in parts) or parts[-1].count("L") < np_


def check_object(obj, want):
    """want: {substring of the mangled name: (callable(name) -> NP, L-groups that may stay in flight)}.
    Returns (rows, errors): rows = (kernel, address, N, NP, histories, slack) with slack = min over the paths of
    (instructions younger than the needed copy) - N: 0 = exactly as counted, > 0 = an over-wait, < 0 = an error."""
    fns = functions(device_disassembly(obj))
    rows, errors = [], []
    for name, ins in sorted(fns.items()):
        np_ = groups = None
        for sub, (fn, g) in want.items():
            if sub in name:
                np_, groups = fn(name), g
        if not np_:
            continue
        waits = analyse(ins, depth=(groups + 1) * np_ + 8 + (20 if groups > 1 else 0))   # the wait's reach + room for extras
        if not waits:
            errors.append("%s: no counted wait found (the pipeline no longer compiles to vmcnt(N) + s_barrier?)" % name)
        for a, n, hists in waits:
            slack = None
            for h in hists:
                if short_weight_copy(h, np_):
                    errors.append("%s @%x: a weight copy of other than NP = %d LDS-DMA instructions on issue history %s" % (name, a, np_, h.lstrip(".")))
                y = younger_than_needed(h, groups * np_ + 1)
                if y is not None:
                    slack = y - n if slack is None else min(slack, y - n)
            rows.append((name, a, n, np_, hists, slack))
            if slack is not None and slack < 0:
                errors.append("%s @%x: s_waitcnt vmcnt(%d) (NP = %d) leaves the needed weight copy in flight on an issue history among %s" % (
                    name, a, n, np_, [h.lstrip(".") for h in hists]))
    return rows, errors


def _np_of_conv3x3(name):   # conv3x3_ragged_kernel<BN, TW, PH, PW, FLAT, SPLIT>: the last template argument is NP (0 = exact)
    m = re.search(r"ELb[01]ELi(\d)EEEv", name)
    return int(m.group(1)) if m else 0


def _np_of_template(name):
    m = re.search(r"kernelILi(\d)EEEv", name)
    return int(m.group(1)) if m else 0


# split::pipeline: the two youngest weight chunks may stay in flight; conv12_fused_split: the tap just requested
TARGETS = {"kernels_rec.o": {"conv3x3_ragged_kernel": (_np_of_conv3x3, 2), "conv12_fused_split_kernel": (_np_of_template, 1)},
           "kernels_nn.o": {"gemm_split_kernel": (_np_of_template, 2)}}


def check_all(build_dir=BUILD):
    rows, errors = [], []
    for obj, want in TARGETS.items():
        r, e = check_object(os.path.join(build_dir, obj), want)
        rows += r
        errors += e
    return rows, errors


if __name__ == "__main__":
    rows, errors = check_all()
    for name, a, n, np_, hists, slack in rows:
        print("%-112s @%06x vmcnt(%2d) NP=%d slack %s" % (name[:112], a, n, np_, "drained" if slack is None else slack))
    over = [r for r in rows if r[5]]
    print("%d counted waits in %d kernels: %d exactly as counted or behind a drain, %d over-waits, %d errors" % (
        len(rows), len({r[0] for r in rows}), len(rows) - len(over), len([r for r in over if r[5] > 0]), len(errors)))
    for e in errors:
        print("ERROR", e)
    sys.exit(1 if errors else 0)
