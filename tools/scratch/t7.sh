set -e
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "se_gate or dw or x3d or X3D or slowfast or uniformer or s3d" > gpurun_out/t7.log 2>&1 || { tail -40 gpurun_out/t7.log; exit 1; }
tail -1 gpurun_out/t7.log
A="MSPI_DW_TILE3=0" B="X=1" bash tools/scratch/ab.sh
