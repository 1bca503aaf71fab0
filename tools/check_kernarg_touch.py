#!/usr/bin/env python3
"""Build-time check of ScalarTouch (csrc/common.h): the kernel-argument prefetch issues `s_load_dword` from inline asm into
"=s" outputs and only waits in done().  The compiler does not know the SGPR write is still in flight: if it redefined one of those
destination registers between issue() and the `s_waitcnt lgkmcnt(0)` of done(), the late load would overwrite the new value.
This compiles a source to gfx950 assembly and verifies, for every kernel, that no instruction between the inline-asm `s_load_dword`
and the inline-asm `s_waitcnt lgkmcnt(0)` writes a destination SGPR of the touch.
usage: python tools/check_kernarg_touch.py ayolov2_amd/csrc/conv.hip [more.hip ...]      (cross-compiles, no GPU needed)"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wno-unused-result",
         "--cuda-device-only", "-S"]


def sregs(tok):
    """SGPR numbers named by an operand token: s5 -> {5}; s[4:7] -> {4, 5, 6, 7}."""
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check(path):
    asm = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [path, "-o", "-"], capture_output=True, text=True)
    if asm.returncode != 0:
        print(asm.stderr[-2000:])
        return 1
    bad = kernels = touched = 0
    name, in_asm, pending, cur_block = None, False, set(), []
    for line in asm.stdout.splitlines():
        t = line.strip()
        m = re.match(r"^(_Z\w+|k_\w+):", t)
        if m:
            name, pending = m.group(1), set()
            kernels += 1
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm, cur_block = True, []
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            for ins in cur_block:
                mm = re.match(r"s_load_dword\s+(s\d+),", ins)
                if mm:
                    if not pending:
                        touched += 1
                    pending |= sregs(mm.group(1))
                if ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins:
                    pending = set()
            continue
        if in_asm:
            cur_block.append(t)
            continue
        if not pending or not t or t.startswith((";", ".", "//")):
            continue
        if re.match(r"s_waitcnt\b.*lgkmcnt\(0\)", t):      # the compiler's own full scalar wait also ends the hazard
            pending = set()
            continue
        ops = re.split(r"[\s,]+", t.split(";")[0].strip())
        if len(ops) >= 2 and ops[0].startswith(("s_", "v_readfirstlane", "v_readlane", "v_cmp")):
            dst = sregs(ops[1])
            if dst & pending:
                bad += 1
                print(f"{path}: {name}: `{t.split(';')[0].strip()}` redefines a touch destination {sorted(dst & pending)} before the wait")
    print(f"{path}: {kernels} functions, {touched} touch sequences (kernel arguments / job structs), {bad} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(max(check(p) for p in sys.argv[1:]))
