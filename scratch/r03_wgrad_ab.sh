#!/bin/bash
for cfg in "0 0" "128 128" "128 64"; do
  set -- $cfg
  ALI_WBM=$1 ALI_WBN=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); r=d['roofline']
print('wtile $1 x $2', d['ms_per_step'], {k:(v['launches'],v['ms'],v['tflops']) for k,v in r['families'].items() if 'wgrad' in k})"
done
