#!/bin/bash
# Is the GPU ever idle inside a train step?  rocprofv3 --kernel-trace of a train-only run; union of all kernel intervals vs wall
# time over the steady-state steps, per HSA queue (main stream / side stream), and the histogram of the gaps between consecutive
# kernels of the busiest queue.   usage (GPU box, repo root): bash tools/gpu_idle.sh
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_gi
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gi -o gi -- python $root/bench.py --steps 12 --warmup 4 --no-extras > $out/gi_bench.json 2> $out/gi.err
f=$(find /tmp/prof_gi -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]) for r in rows))
# steady state: from the 7th to the 15th k_pack_input (one per train step; 4 warm-up + 12 timed steps, then the in-situ measurement runs)
packs = [s for s, e, q, n in ks if "k_pack_input" in n]
lo, hi = packs[6], packs[14]
nsteps = 8
ks = [k for k in ks if lo <= k[0] < hi]
wall = hi - lo
print(f"{nsteps} steps, {wall / nsteps / 1e6:.3f} ms per step under the tracer")
def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    return tot + (cur_e - cur_s if cur_e is not None else 0)
allu = union([(s, e) for s, e, q, n in ks])
print(f"window {wall/1e6:.2f} ms: some kernel running {allu/1e6:.2f} ms = {allu/wall*100:.1f} %; idle {(wall-allu)/1e6:.2f} ms")
byq = collections.defaultdict(list)
for s, e, q, n in ks: byq[q].append((s, e, n))
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    u = union([(s, e) for s, e, n in v])
    print(f"queue {q}: {len(v)} kernels, busy {u/1e6:.2f} ms = {u/wall*100:.1f} % of the window")
q0 = max(byq.items(), key=lambda kv: len(kv[1]))[1]
gaps = [q0[i + 1][0] - q0[i][1] for i in range(len(q0) - 1)]
import statistics
pos = [g for g in gaps if g > 0]
print(f"busiest queue: {len(gaps)} gaps, median {statistics.median(gaps)/1e3:.2f} us, mean of positive {sum(pos)/max(len(pos),1)/1e3:.2f} us, sum of positive {sum(pos)/1e6:.2f} ms")
for lim in (2e3, 5e3, 10e3, 50e3, 1e9):
    sel = [g for g in pos if g <= lim]
    print(f"  gaps <= {lim/1e3:.0f} us: {len(sel)} totalling {sum(sel)/1e6:.3f} ms")
big = sorted(((q0[i + 1][0] - q0[i][1], q0[i][2][:50], q0[i + 1][2][:50]) for i in range(len(q0) - 1)), reverse=True)[:12]
for g, a, b in big: print(f"  {g/1e3:8.1f} us between {a} -> {b}")
PY
