# Runs on the GPU box: everything profiles/<round>/ holds - rocprofv3 kernel trace + PMC passes of the headline in both
# NMS modes, of config 4, of config 1 (cross-class) and of the headline at 13 % box coverage, then the bench lines.
set -e
bash tools/gpu_profile.sh r03_pc per_class > /dev/null
bash tools/gpu_profile.sh r03_cc cc > /dev/null
bash tools/gpu_profile.sh r03_c4 per_class --config 4 > /dev/null
bash tools/gpu_profile.sh r03_c1 cc --config 1 > /dev/null
bash tools/gpu_profile.sh r03_cov per_class --box-coverage 0.13 > /dev/null
bash tools/gpu_profile.sh r03_kept per_class --kept-outputs > /dev/null
bash tools/bench_all.sh
timeout -k 10 120 python3 tools/bench_consumer.py > gpurun_out/ball/consumer.json 2> gpurun_out/ball/consumer.err
timeout -k 10 600 python3 bench.py > gpurun_out/ball/default.json 2> gpurun_out/ball/default.err
