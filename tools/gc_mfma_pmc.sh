set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/gcm
export YOLACT_POST_LIB=$PWD/yolact_amd/libyp_gcmfma.so
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/gcm/pmc_mfma -- python3 tools/f4_backward_loop.py > gpurun_out/gcm/pmc_mfma.log 2>&1
tail -3 gpurun_out/gcm/pmc_mfma.log
