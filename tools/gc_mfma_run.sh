set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/gcm
timeout -k 10 150 python3 tools/gc_mfma_check.py run gpurun_out/gcm/default.npy > gpurun_out/gcm/log.txt 2>&1
export YOLACT_POST_LIB=$PWD/yolact_amd/libyp_gcmfma.so
timeout -k 10 150 python3 tools/gc_mfma_check.py run gpurun_out/gcm/mfma.npy >> gpurun_out/gcm/log.txt 2>&1
timeout -k 10 60 python3 tools/gc_mfma_check.py cmp gpurun_out/gcm/default.npy gpurun_out/gcm/mfma.npy >> gpurun_out/gcm/log.txt 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gcm/trace_mfma -- python3 tools/f4_backward_loop.py > gpurun_out/gcm/trace_mfma.log 2>&1
unset YOLACT_POST_LIB
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gcm/trace_default -- python3 tools/f4_backward_loop.py > gpurun_out/gcm/trace_default.log 2>&1
cat gpurun_out/gcm/log.txt
