"""Register / LDS / occupancy figures of the library's kernels, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/kernel_resources.py [substring ...]   (no GPU needed)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pats = sys.argv[1:] or ["smc_step"]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DCPPROB_HIP_BUILD", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "cpprob_amd", "include"), "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/cpprob_res.so",
           os.path.join(ROOT, "cpprob_amd", "csrc", "cpprob_hip.hip")]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].strip()
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if not any(p in dn for p in pats):
            continue

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"
        print("%-130s VGPR %s AGPR %s SGPR %s spill %s scratch %s occ %s LDS %s" % (dn[:130], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("VGPR Spill"), g(r"ScratchSize \[bytes/lane\]"),
                                                                                      g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()
