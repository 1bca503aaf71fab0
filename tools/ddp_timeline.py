#!/usr/bin/env python3
"""rocprofv3 kernel trace of `AYOLO_FORCE_DDP=1 bench.py` -> where the RCCL kernels sit relative to backward.
A step's backward window = first k_loss_grad_packed launch .. end of the last k_wgrad / k_bn_bwd_apply before the optimiser's
k_sgd_step.  For every RCCL kernel (name contains 'nccl' / 'rccl') of the step: start and end offset from the window's start,
whether it ends before the window does, and which compute kernels ran concurrently.
usage: python tools/ddp_timeline.py <kernel_trace.csv>"""
import csv
import sys


def short(name):
    name = name[5:] if name.startswith("void ") else name
    return name[:56]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    # the queue that carries most launches is the compute stream; RCCL runs its kernels (a multi-rank group) or its device
    # copies (__amd_rocclr_copyBuffer: what a one-rank group's all-reduce degenerates to) on its own stream / queue
    per_q = {}
    for r in rows:
        per_q.setdefault((r[3], r[4]), []).append(r)
    print("queues / streams in the trace:")
    for q, v in sorted(per_q.items(), key=lambda kv: -len(kv[1])):
        names = {}
        for r in v:
            names[short(r[2])] = names.get(short(r[2]), 0) + 1
        top = sorted(names.items(), key=lambda kv: -kv[1])[:4]
        print(f"  queue {q[0]} stream {q[1]}: {len(v)} launches, {sum(r[1] - r[0] for r in v) / 1e6:.2f} ms busy; {top}")
    main_q = max(per_q.items(), key=lambda kv: len(kv[1]))[0]
    wgrad_q = {(r[3], r[4]) for r in rows if r[2].startswith("_Z7k_wgrad")}
    is_comm = lambda r: ("nccl" in r[2].lower() or "rccl" in r[2].lower()
                         or ((r[3], r[4]) != main_q and (r[3], r[4]) not in wgrad_q and "fillBuffer" not in r[2]))
    sgd = [i for i, r in enumerate(rows) if r[2].startswith("k_sgd_step")]
    print(f"{len(rows)} kernel launches, {len(sgd)} optimiser steps, {sum(is_comm(r) for r in rows)} exchange launches (RCCL kernels / its device copies)")
    done = 0
    for si in sgd[-3:]:                                   # the last three (timed) steps
        t_sgd = rows[si][0]
        j = si
        while j > 0 and "k_loss_grad_packed" not in rows[j][2]:
            j -= 1
        while j > 0 and "k_loss_grad_packed" in rows[j - 1][2]:
            j -= 1
        t0 = rows[j][0]
        bwd = [r for r in rows[j:si] if not is_comm(r)]
        last_compute = max(r[1] for r in bwd if r[2].startswith(("_Z7k_wgrad", "_Z14k_bn_bwd_apply", "_Z7k_gconv", "_Z15k_bn_bwd_reduce")))
        comm = [r for r in rows[j:si + 40] if is_comm(r) and r[0] < t_sgd + 2_000_000 and r[0] >= t0]
        print(f"\nstep ending at optimiser launch #{si}: backward window {(last_compute - t0) / 1e3:.0f} us "
              f"({len(bwd)} compute launches), optimiser starts at +{(t_sgd - t0) / 1e3:.0f} us")
        for c in comm:
            conc = [r for r in bwd if r[0] < c[1] and r[1] > c[0]]
            inside = "inside" if c[1] <= last_compute else ("straddles the end" if c[0] < last_compute else "after")
            names = sorted({short(r[2])[:28] for r in conc})
            print(f"  exchange {short(c[2]):56s} +{(c[0] - t0) / 1e3:8.0f} .. +{(c[1] - t0) / 1e3:8.0f} us  ({(c[1] - c[0]) / 1e3:6.0f} us)  {inside} "
                  f"the backward window; {len(conc)} compute kernels concurrent {names[:4]}")
        done += 1
    if not done:
        print("no optimiser step found in the trace")


if __name__ == "__main__":
    main()
