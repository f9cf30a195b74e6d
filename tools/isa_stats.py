#!/usr/bin/env python3
"""Per-kernel instruction statistics of an AMDGPU assembly listing (hipcc -S --cuda-device-only): instructions, branches, back edges,
LDS / global memory instructions, registers.  Used to check what a compile-time fact removed from a kernel (tools/, tests/test_host.py)."""
import re
import sys


def stats(path):
    out, cur, labels = {}, None, {}
    lines = open(path).read().split("\n")
    for ln, l in enumerate(lines):
        m = re.match(r"^([_A-Za-z][\w$.]*):", l)
        if m and not l.startswith(".LBB") and not l.startswith(".L"):
            cur = m.group(1)
            out[cur] = dict(n=0, branches=0, back_edges=0, lds=0, vmem=0, valu=0, salu=0)
            labels = {}
            continue
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = ln
            continue
        if cur is None or not l.startswith("\t"):
            continue
        tok = l.strip().split()
        if not tok:
            continue
        ins = tok[0]
        if not ins.startswith(("s_", "v_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
            continue
        d = out[cur]
        d["n"] += 1
        if ins.startswith("s_cbranch") or ins == "s_branch":
            d["branches"] += 1
            tgt = tok[1] if len(tok) > 1 else ""
            if tgt in labels:                       # target defined above: a back edge
                d["back_edges"] += 1
        elif ins.startswith("ds_"):
            d["lds"] += 1
        elif ins.startswith(("global_", "buffer_", "flat_", "scratch_")):
            d["vmem"] += 1
        elif ins.startswith("v_"):
            d["valu"] += 1
        else:
            d["salu"] += 1
    return out


if __name__ == "__main__":
    for k, v in stats(sys.argv[1]).items():
        if v["n"]:
            print(v, k[:110])
