set -e
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "x3d or X3D or conv or gate or se_ or dw or swish or sigmoid or mlp" > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
bash tools/scratch/ab.sh
