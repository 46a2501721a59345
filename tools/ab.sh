#!/bin/bash
# A/B of bench.py under environment variants, interleaved in one box session:  bash tools/ab.sh <tag> <rounds> "VAR=1 VAR2=x" "..." ...
TAG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 10 --warmup 3 > $OUT/ab_${i}_$r.json 2> $OUT/ab_${i}_$r.err || { echo "variant $i failed"; tail -5 $OUT/ab_${i}_$r.err; }
    echo "round $r [$v] $(python3 -c "import json;d=json.load(open('$OUT/ab_${i}_$r.json'));print(d['value'], d['ms_per_step'])" 2>/dev/null)"
  done
done
