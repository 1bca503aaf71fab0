#!/usr/bin/env python3
"""Sum of the op times of a tools/op_table.py table on the <= 40 x 40 maps (VERDICT r5 item 1's bar) and on the larger maps, main chain
only (weight-gradient groups excluded).  usage: python tools/small_map_sum.py table.txt [...]"""
import re
import sys

for path in sys.argv[1:]:
    small = big = 0.0
    ns = nb = 0
    bytes_s = bytes_b = 0.0
    for line in open(path):
        if line.startswith("---- families"):
            break
        m = re.search(r"^\s*(\d+)\s+(OP_\w+)\s+(.*?)\s+([\d.]+) us\s+([\d.]+) MB", line)
        if not m:
            continue
        _, op, desc, us, mb = m.groups()
        us, mb = float(us), float(mb)
        if us <= 0 or op in ("OP_WGRAD_GROUP", "OP_STEM_BN_WGRAD", "OP_CONV_WGRAD"):
            continue
        hw = None
        mm = re.search(r"(\d+)->\s*(\d+)\s*$", desc)            # conv: "... 80-> 40": the larger side of the op
        if mm:
            hw = max(int(mm.group(1)), int(mm.group(2)))
            if op == "OP_CONV_FWD" or op == "OP_CONV_DGRAD":
                hw = int(mm.group(2)) if op == "OP_CONV_FWD" else int(mm.group(2))
        mm = re.search(r"npix=(\d+)", desc)
        if mm:
            hw = int(round((int(mm.group(1)) / 64) ** 0.5))
        if hw is None:
            # pools / upsampling / copies: by bytes (a 40 x 40 x 512 x 64 fp16 map is 105 MB; everything below 2 x that is small)
            hw = 40 if mb < 220 else 80
        if hw <= 40:
            small += us; ns += 1; bytes_s += mb
        else:
            big += us; nb += 1; bytes_b += mb
    print(f"{path}: <=40^2: {ns} ops {small / 1e3:.3f} ms {bytes_s / 1e3:.2f} GB ({bytes_s / max(small, 1e-9):.2f} TB/s) | larger: {nb} ops {big / 1e3:.3f} ms "
          f"{bytes_b / 1e3:.2f} GB ({bytes_b / max(big, 1e-9):.2f} TB/s)")
