# Runs on the GPU box: the randomised parity fuzzes and the run-to-run soak, in pieces that each report within minutes.
set -e
mkdir -p gpurun_out/fuzz
for s in 11 12 13; do timeout -k 10 400 python3 -u tools/fuzz_vs_oracle.py 500 $s 2>&1 | tail -5 | tee -a gpurun_out/fuzz/detect.txt; done
for s in 11 12; do timeout -k 10 400 python3 -u tools/fuzz_rle.py 800 $s 2>&1 | tail -5 | tee -a gpurun_out/fuzz/rle.txt; done
timeout -k 10 400 python3 -u tools/soak.py 100 2 2>&1 | tail -5 | tee -a gpurun_out/fuzz/soak.txt
timeout -k 10 400 python3 -u tools/soak.py 40 4 2>&1 | tail -5 | tee -a gpurun_out/fuzz/soak.txt
