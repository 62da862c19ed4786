#!/bin/bash
# Runs on the GPU box (via gpurun): score_filter time against tile size (rebuilds) and resident waves per CU.
# usage: tools/sf_sweep.sh "<ta> ..." "<waves> ..."
mkdir -p gpurun_out
for ta in $1; do
  timeout -k 10 300 python -m yolact_amd.build -DYP_SF_TA=$ta > /dev/null 2>&1 || exit 1
  for w in $2; do
    YP_TUNE_SF_WAVES=$w timeout -k 10 120 python bench.py --config 2 --no-cpu-baseline --skip-same-buffer > gpurun_out/sw.json 2> gpurun_out/sw.err || exit 1
    python -c "
import json
d=json.load(open('gpurun_out/sw.json')); print('ta=$ta waves=$w', d['value'], d['kernels_ms'])"
  done
done
