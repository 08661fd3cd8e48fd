#!/bin/bash
# A/B of library variants on the bench's timed region: tools/ab_bench.sh <lib or ""> ...  (alternating, two rounds)
for round in 1 2; do
  for lib in "$@"; do
    v=$(GPSCAL_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-track --no-loam --no-single-pair 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f it/s  %.3f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
    echo "round $round lib '${lib}': $v"
  done
done
