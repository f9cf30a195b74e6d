#!/usr/bin/env python3
"""Static instruction budget of a kernel, part by part: the library compiled with -DCPPROB_MARKS leaves a comment line in its
assembly at every CPH_STAMP(k); this counts the instructions between consecutive marks (in program order: a loop's body counts once,
and the scheduler may have moved a few ALU instructions across a mark).
usage: python tools/isa_budget.py <substring of the demangled kernel name> ...      (no GPU needed; ~2 min of hipcc)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pats = sys.argv[1:] or ["smc_step_counts_kernel<cph::ModelHmm3, false, 0>"]
    out = "/tmp/cpprob_marks.s"
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-DCPPROB_HIP_BUILD", "-DCPPROB_MARKS", "-DCPPROB_BUILD_ID=\"marks\"", "--cuda-device-only", "-S",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cpprob_amd", "include"), "-o", out, os.path.join(ROOT, "cpprob_amd", "csrc", "cpprob_hip.hip")]
    subprocess.check_call(cmd)
    cur, parts = None, {}
    for l in open(out):
        m = re.match(r"^(_Z[\w$.]+):", l)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            parts[cur] = [["entry", dict(valu=0, salu=0, lds=0, vmem=0, branch=0)]]
            continue
        if cur is None:
            continue
        m = re.search(r"; CPH_MARK (\d+)", l)
        if m:
            parts[cur].append(["mark %s" % m.group(1), dict(valu=0, salu=0, lds=0, vmem=0, branch=0)])
            continue
        if not l.startswith("\t"):
            continue
        tok = l.strip().split()
        if not tok:
            continue
        ins = tok[0]
        d = parts[cur][-1][1]
        if ins.startswith("s_cbranch") or ins == "s_branch":
            d["branch"] += 1
        elif ins.startswith("v_"):
            d["valu"] += 1
        elif ins.startswith("ds_"):
            d["lds"] += 1
        elif ins.startswith(("global_", "buffer_", "flat_", "scratch_")):
            d["vmem"] += 1
        elif ins.startswith("s_"):
            d["salu"] += 1
    for name, ps in parts.items():
        if not any(p in name for p in pats) or len(ps) < 2:
            continue
        print(name[:160])
        tot = dict(valu=0, salu=0, lds=0, vmem=0, branch=0)
        for label, d in ps:
            print("  from %-8s  valu %4d  salu %4d  lds %3d  vmem %3d  branch %3d" % (label, d["valu"], d["salu"], d["lds"], d["vmem"], d["branch"]))
            for k in tot:
                tot[k] += d[k]
        print("  total          valu %4d  salu %4d  lds %3d  vmem %3d  branch %3d" % (tot["valu"], tot["salu"], tot["lds"], tot["vmem"], tot["branch"]))


if __name__ == "__main__":
    main()
