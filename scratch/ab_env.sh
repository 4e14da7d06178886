#!/bin/bash
# A = default (16x16x32), B = FACL_GEMM_MFMA32=1 (old shape), alternating, same box
for i in 1 2 3; do
  for v in new old; do
    unset FACL_GEMM_MFMA32
    [ $v = old ] && export FACL_GEMM_MFMA32=1
    timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=[d['roofline']]+d['roofline_more'];f={x['kernel']:x['ms_per_launch'] for x in r if 'gemm' in x['kernel'] and '49152' in x['kernel']};print('$v',d['ms_per_step'],d['final_loss'],' '.join(f'{v:.4f}' for k,v in sorted(f.items())), round(sum(f.values()),4))"
  done
done
