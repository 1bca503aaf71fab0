#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_sq.sh r01
# one --pmc pass of the 8 SQ slots + GRBM_GUI_ACTIVE over a train-only run (no trace domains besides --kernel-trace):
# MFMA busy share, wave-cycle breakdown (parked / issue-stalled / issuing) and LDS bank-conflict share per kernel family.
tag=${1:-r01}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --output-format csv -d /tmp/prof_sq -o sq -- python $root/bench.py --steps 4 --warmup 2 --no-extras > /dev/null 2> $out/${tag}_pmc_sq.err
python $root/tools/pmc_sq_summary.py $(find /tmp/prof_sq -name '*counter_collection.csv' | head -1) 6 $out/${tag}_pmc_sq.json | tee $out/${tag}_pmc_sq.txt
tail -2 $out/${tag}_pmc_sq.err
