#!/bin/bash
# Runs on the GPU box: bench config 2 with the library built with the given extra flags, one build per argument.
mkdir -p gpurun_out
for fl in "$@"; do
  timeout -k 10 300 python -m yolact_amd.build $fl > /dev/null 2>&1 || exit 1
  timeout -k 10 120 python bench.py --config 2 --no-cpu-baseline --skip-same-buffer > gpurun_out/sw.json 2> gpurun_out/sw.err || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/sw.json')); print('$fl', d['value'], d['kernels_ms'], d.get('parity',{}).get('vs_c_oracle_same_contract'))"
done
