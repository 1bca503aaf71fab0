#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_sq_cmd.sh <tag> <runs> <command...>
# Two --pmc passes over <command> (no trace domain besides --kernel-trace):
#   pass 1: the 8 SQ slots of tools/pmc_sq.sh (MFMA busy share, wave-cycle breakdown, LDS conflicts) + GRBM_GUI_ACTIVE
#   pass 2: the issue split -- cycles with a VALU / scalar / LDS / vector-memory / MISC instruction issuing, instruction counts of
#           VALU, MFMA, SALU, LDS -- restricted to the counters this rocprofv3 lists (-L)
# <runs> = how many times <command> executes the workload (divisor for the per-run columns).
tag=$1; runs=$2; shift 2
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/psq1 /tmp/psq2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --output-format csv -d /tmp/psq1 -o sq -- "$@" > /dev/null 2> $out/${tag}_pmc_sq.err
python $root/tools/pmc_sq_summary.py $(find /tmp/psq1 -name '*counter_collection.csv' | head -1) $runs $out/${tag}_pmc_sq.json | tee $out/${tag}_pmc_sq.txt
avail=$(rocprofv3 -L 2>/dev/null | tr -c 'A-Za-z0-9_' '\n' | sort -u)
want="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"
have=""
n=0
for c in $want; do
    if echo "$avail" | grep -qx "$c"; then have="$have $c"; n=$((n + 1)); fi
    [ $n -ge 8 ] && break
done
echo "pass 2 counters:$have" | tee -a $out/${tag}_pmc_sq.txt
if [ -n "$have" ]; then
    rocprofv3 --kernel-trace --pmc $have GRBM_GUI_ACTIVE --output-format csv -d /tmp/psq2 -o sq2 -- "$@" > /dev/null 2>> $out/${tag}_pmc_sq.err
    python - $(find /tmp/psq2 -name '*counter_collection.csv' | head -1) $runs >> $out/${tag}_pmc_sq.txt <<'PY'
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
runs = float(sys.argv[2])
print("issue split per kernel (share of wave quad-cycles with an instruction of that type issuing; instruction counts per run, millions)")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0))[:14]:
    w = c.get("SQ_WAVE_CYCLES", 0.0)
    if w <= 0:
        continue
    sh = {n.replace("SQ_ACTIVE_INST_", "").lower(): round(v / w, 3) for n, v in c.items() if n.startswith("SQ_ACTIVE_INST_")}
    cnt = {n.replace("SQ_INSTS_", "").lower(): round(v / runs / 1e6, 2) for n, v in c.items() if n.startswith("SQ_INSTS_")}
    print(f"{k[:58]:58s} issuing {sh}  insts_M {cnt}")
PY
fi
tail -2 $out/${tag}_pmc_sq.err
tail -16 $out/${tag}_pmc_sq.txt
