#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel of a HIP source, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/kernel_resources.py ayolov2_amd/csrc/conv.hip [filter-substring]   (cross-compiles, no GPU needed)
       python tools/kernel_resources.py --remarks FILE [filter]                        (a saved remark log)"""
import re
import subprocess
import sys

args = sys.argv[1:]
if args[0] == "--remarks":
    txt = open(args[1]).read()
    args = args[1:]
else:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
           "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage", "-c", args[0], "-o", "/dev/null"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
flt = args[1] if len(args) > 1 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
K_SCR, K_OCC, K_LDS = r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"


def field(b, k):
    m = re.search(k + r": (\d+)", b)
    return int(m.group(1)) if m else -1


print(f"{'VGPR':>5}{'AGPR':>5}{'scr':>5}{'SGPR':>5}{'occ':>4}{'LDS':>7}  kernel")
for b, d in zip(blocks, dem):
    d = re.sub(r"^void ", "", d)
    d = re.sub(r"\(GConvP\)|\(WGradP\)", "", d).replace("_Float16", "f16")
    if flt and flt not in d:
        continue
    print(f"{field(b, 'VGPRs'):5d}{field(b, 'AGPRs'):5d}{field(b, K_SCR):5d}{field(b, 'SGPRs'):5d}{field(b, K_OCC):4d}{field(b, K_LDS):7d}  {d[:100]}")
