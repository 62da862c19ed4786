# Runs on the GPU box: the bench lines of every BASELINE config (no CPU baseline), one JSON file each under gpurun_out/ball/.
set -e
mkdir -p gpurun_out/ball
run() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --skip-same-buffer --steps 200 --warmup 20 "$@" > gpurun_out/ball/$name.json 2> gpurun_out/ball/$name.err; }
run c2
run c2cc --nms-mode cc
run c4 --config 4
run c3 --config 3
run c5 --config 5
run c1cc --config 1 --nms-mode cc
run c2cov --box-coverage 0.13
run c2kept --kept-outputs
run c2ref --reference-layout
run c2refkept --reference-layout --kept-outputs
