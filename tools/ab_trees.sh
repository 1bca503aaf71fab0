#!/bin/bash
# A/B of whole TREES on the same box (a baseline checkout under ab/<name>/ with its own built library against this tree, or
# this tree under different environment switches).  usage (GPU box, repo root):
#   bash tools/ab_trees.sh <rounds> <steps> "label=ENV=.. ENV=.. path/to/bench.py" ...
# Alternates the arms <rounds> times; prints ms_per_step of every run and the per-arm median.
rounds=$1; steps=$2; shift 2
declare -A all
for i in $(seq 1 $rounds); do
  for arm in "$@"; do
    label=${arm%%=*}; cmd=${arm#*=}
    ms=$(env $cmd --no-extras --steps $steps --warmup 5 2>/tmp/ab_err.txt | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    [ -z "$ms" ] && { echo "$label FAILED"; tail -5 /tmp/ab_err.txt; ms=nan; }
    echo "round $i  $label  $ms ms/step"
    all[$label]="${all[$label]} $ms"
  done
done
for label in "${!all[@]}"; do
  python - "$label" ${all[$label]} <<'PY'
import sys, statistics
v = [float(x) for x in sys.argv[2:] if x != "nan"]
print("median  %-28s %.3f ms/step  (n=%d, min %.3f max %.3f)" % (sys.argv[1], statistics.median(v) if v else float("nan"), len(v), min(v) if v else 0, max(v) if v else 0))
PY
done
