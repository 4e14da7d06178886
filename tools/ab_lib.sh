#!/bin/bash
# same-box A/B of two builds of the library: alternating bench runs (FACL_LIB selects the build)
# usage: tools/ab_lib.sh [prev.so] [rounds] [kernel-name substring whose in-step time is printed as well]
A=${1:-scratch/lib_prev.so}; N=${2:-3}; K=${3:-}
for i in $(seq $N); do
  for L in "$A" ""; do
    if [ -n "$L" ]; then export FACL_LIB=$PWD/$L; tag=prev; else unset FACL_LIB; tag=new; fi
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k='$K'
ks=[(r['kernel'], r['ms_per_launch']) for r in [d['roofline']]+d['roofline_more'] if k and k in r['kernel']]
print('$tag', d['ms_per_step'], d['ms_per_step_median_fenced'], ks)"
  done
done
