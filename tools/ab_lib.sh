#!/bin/bash
# same-box A/B of two builds of the library: alternating bench runs (FACL_LIB selects the build)
A=${1:-scratch/lib_prev.so}; N=${2:-3}
for i in $(seq $N); do
  for L in "$A" ""; do
    if [ -n "$L" ]; then export FACL_LIB=$PWD/$L; tag=prev; else unset FACL_LIB; tag=new; fi
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['ms_per_step_median_fenced'])"
  done
done
