set -e
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "se_fold or se_gate or dwconv_pool" > gpurun_out/t2.log 2>&1 || { tail -40 gpurun_out/t2.log; exit 1; }
tail -3 gpurun_out/t2.log
