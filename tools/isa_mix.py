"""Instruction mix of the loops of a kernel in a `hipcc -S` listing (the tile loops of attention.hip: what a tile costs in
issue slots, by class).  python tools/isa_mix.py attention.s attn_fwd_kernelILb1ELb1 [min_loop_lines]"""
import collections
import re
import sys

CLASSES = [("mfma", ("v_mfma",)), ("trans", ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")), ("v_pk", ("v_pk_",)),
           ("cvt_pk", ("v_cvt_pk",)), ("mul24", ("v_mul_u32_u24", "v_mad_u32_u24")), ("cmp", ("v_cmp",)),
           ("cndmask", ("v_cndmask",)), ("accvgpr", ("v_accvgpr",)), ("v_mov", ("v_mov",)), ("valu_other", ("v_",)),
           ("ds_read", ("ds_read", "ds_load")), ("ds_write", ("ds_write", "ds_store")),
           ("vmem_ld", ("global_load", "buffer_load")), ("vmem_st", ("global_store", "buffer_store", "global_atomic", "buffer_atomic")),
           ("waitcnt", ("s_waitcnt",)), ("s_nop", ("s_nop",)), ("barrier", ("s_barrier",)), ("salu", ("s_",))]


def classify(op):
    for name, prefixes in CLASSES:
        if op.startswith(prefixes):
            return name
    return op


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2]
    min_lines = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(want) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    labels = {}
    for i in range(start, end):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            labels[m.group(1)] = i
    print(lines[start].split(":")[0], end - start, "lines")
    for i in range(start, end):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if not m or m.group(1) not in labels or labels[m.group(1)] >= i or i - labels[m.group(1)] < min_lines:
            continue
        a = labels[m.group(1)]
        mix, other, nops = collections.Counter(), collections.Counter(), 0
        for l in lines[a:i + 1]:
            l = l.strip()
            if not l or l[0] in ".;/" or l.endswith(":"):
                continue
            op = l.split()[0]
            k = classify(op)
            mix[k] += 1
            if k == "valu_other":
                other[op] += 1
            if k == "s_nop":
                nops += int(l.split()[1]) + 1
        print(f"  loop lines {a}-{i}:", dict(mix.most_common()), "s_nop wait states", nops)
        print("     other VALU:", dict(other.most_common(16)))


if __name__ == "__main__":
    main()
