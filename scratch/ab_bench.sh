#!/bin/bash
# usage: scratch/ab_bench.sh <kernel-prefix>   (A = in-tree lib, B = scratch/lib_old.so), alternating, same box
for i in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export FACL_LIB=$PWD/scratch/lib_old.so; else unset FACL_LIB; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=[d['roofline']]+d['roofline_more'];f=[x for x in r if x['kernel'].startswith('$1')][0];print('$v',d['ms_per_step'],f['ms_per_launch'])"
  done
done
