#!/bin/bash
# A/B: XCD-contiguous tile order vs raster on the spectrogram benches
set -e -o pipefail
for w in "esrf f16" "esrf f32" "audio f16" "audio f32"; do
  set -- $w
  for x in 0 1; do
    ALI_NO_XCD=$x python bench.py --workload $1 --precision $2 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('$1 $2 no_xcd=$x', d['ms_per_step'], d['value'], r['achieved'], {k:(v['launches'],v['ms'],v['tflops']) for k,v in r['families'].items()})"
  done
done
