#!/usr/bin/env python3
"""Instruction mix of a kernel's hot loop from hipcc's device assembly ("an ISA-level count, per k-step of
the dominant variant, of VALU / SALU / ds_read / buffer_load issue slots against the MFMA cycles").
usage: hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -o conv.s ayolov2_amd/csrc/conv.hip
       python tools/isa_loop_count.py conv.s <mangled kernel name>
Prints, for every basic block of the kernel that contains MFMAs, the count of instructions by class; the step loop's body is the
block (or chain of blocks) with the step's 16 / 8 MFMAs and its s_barrier."""
import collections
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read"
    if op.startswith("ds_"):
        return "ds_write"
    if op.startswith("buffer_load") or op.startswith("global_load"):
        return "vmem_load"
    if op.startswith("buffer_store") or op.startswith("global_store") or op.startswith("buffer_atomic") or op.startswith("global_atomic"):
        return "vmem_store"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_rsq") or op.startswith("v_sqrt") or op.startswith("v_log"):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu"
    return "other"


blocks, cur, name = [], collections.Counter(), "entry"
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append((name, cur))
            cur, name = collections.Counter(), t.split(":")[0]
        continue
    if re.match(r"^[A-Za-z_.$][\w.$]*:", t):
        continue
    op = t.split()[0]
    cur[cls(op)] += 1
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        cur["->" + t.split()[-1]] += 0
blocks.append((name, cur))
order = ["mfma", "valu", "valu_trans", "salu", "smem", "ds_read", "ds_write", "vmem_load", "vmem_store", "s_waitcnt", "s_barrier", "s_nop", "branch"]
print(f"{'block':14s} " + " ".join(f"{o:>10s}" for o in order))
tot = collections.Counter()
for name, c in blocks:
    for k, v in c.items():
        tot[k] += v
    if c["mfma"] == 0 and c["s_barrier"] == 0:
        continue
    print(f"{name:14s} " + " ".join(f"{c[o]:10d}" for o in order))
print(f"{'whole kernel':14s} " + " ".join(f"{tot[o]:10d}" for o in order))
