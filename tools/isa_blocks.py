"""Instruction-class histogram per basic block of one kernel in a hipcc -S listing.

    python tools/isa_blocks.py conv.s _Z24conv3x3_halo_gemm_kernelILi0ELb0ELi1ELb1EEv8ConvArgs [min_mfma]
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, label = [], Counter(), "entry"
    ops = Counter()
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((label, cur, ops))
            label, cur, ops = m.group(1), Counter(), Counter()
            continue
        s = l.strip()
        if not s or s.startswith((";", ".")):
            continue
        op = s.split()[0]
        cur[classify(op)] += 1
        if classify(op) in ("valu", "salu"):
            ops[op] += 1
    blocks.append((label, cur, ops))
    total = Counter()
    for label, c, o in blocks:
        total.update(c)
        if c["mfma"] >= min_mfma:
            print(label, dict(c))
            print("    ", o.most_common(14))
    print("TOTAL", dict(total))


main()
