#!/usr/bin/env python
"""Registers / LDS / occupancy of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py egonn_amd/csrc/sconv.hip [name-filter]"""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-c", "-Rpass-analysis=kernel-resource-usage",
                        "-I" + os.path.join(REPO, "egonn_amd", "csrc"), "-I" + os.path.join(REPO, "include"), src, "-o", os.path.join(d, "o.o")],
                       capture_output=True, text=True)
t = r.stderr
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void egonn::", "")
    if flt and flt not in dem:
        continue
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-48s VGPR %3s AGPR %3s SGPR %3s occ %s LDS %6s scratch %s" % (dem, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"Occupancy \[waves/SIMD\]"),
                                                                       g(r"LDS Size \[bytes/block\]"), g(r"ScratchSize \[bytes/lane\]")))
