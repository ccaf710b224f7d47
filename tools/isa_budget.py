#!/usr/bin/env python3
"""isa_budget.py FILE.s KERNEL_SUBSTRING: static instruction counts of one kernel of a `hipcc -S --cuda-device-only` listing,
per stretch between two s_barrier instructions (in program order) and per basic block: VALU / SALU / LDS / VMEM / other.
Dynamic counts = these x trip counts (tools/../profiles/r03_tile_budget.md does that by hand and checks the sum against
SQ_INSTS_VALU / SQ_INSTS_SALU)."""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep") or op.startswith(("s_branch", "s_cbranch", "s_setprio")):
        return "ctl"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    per_block = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_.$0-9]+:", l) and key in l and not l.startswith(".L"):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    seg, segs, blocks = {}, [], []
    cur_block, cur = None, {}
    n_barrier = 0
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            if cur_block is not None:
                blocks.append((cur_block, cur))
            cur_block, cur = m.group(1), {}
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        seg[c] = seg.get(c, 0) + 1
        cur[c] = cur.get(c, 0) + 1
        if op == "s_barrier":
            segs.append((n_barrier, seg))
            seg = {}
            n_barrier += 1
    segs.append((n_barrier, seg))
    if cur_block is not None:
        blocks.append((cur_block, cur))
    keys = ["valu", "salu", "lds", "vmem", "smem", "ctl", "other"]
    print("stretch  " + "  ".join("%6s" % k for k in keys))
    tot = {k: 0 for k in keys}
    for i, sg in segs:
        print("%-8d " % i + "  ".join("%6d" % sg.get(k, 0) for k in keys))
        for k in keys:
            tot[k] += sg.get(k, 0)
    print("total    " + "  ".join("%6d" % tot[k] for k in keys))
    if per_block:
        print()
        for name, b in blocks:
            print("%-14s " % name + "  ".join("%6d" % b.get(k, 0) for k in keys))


if __name__ == "__main__":
    main()
