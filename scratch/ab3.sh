#!/bin/bash
for i in 1 2; do
  for v in new new4 old; do
    unset FACL_LIB FACL_FWD3_W4
    [ $v = old ] && export FACL_LIB=$PWD/scratch/lib_old.so
    [ $v = new4 ] && export FACL_FWD3_W4=1
    timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=[d['roofline']]+d['roofline_more'];f=[x for x in r if x['kernel'].startswith('k_sa_fwd3')][0];print('$v',d['ms_per_step'],f['ms_per_launch'],d['final_loss'])"
  done
done
