#!/bin/bash
mkdir -p gpurun_out
for x in $1; do
  timeout -k 10 300 python -m yolact_amd.build -DYP_SF_X=$x > /dev/null 2>&1 || exit 1
  timeout -k 10 120 python bench.py --config 2 --no-cpu-baseline --skip-same-buffer > gpurun_out/sw.json 2> gpurun_out/sw.err || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/sw.json')); print('x=$x', d['value'], d['kernels_ms'])"
done
