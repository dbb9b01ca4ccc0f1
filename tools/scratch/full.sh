set -e
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu > gpurun_out/full_gpu.log 2>&1 || { tail -40 gpurun_out/full_gpu.log; exit 1; }
tail -2 gpurun_out/full_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
