# same-box A/B of the bench line: base library (tools/scratch/libmspi_hip_base.so) vs the tree's
set -e
mkdir -p gpurun_out
run() { tag=$1; shift; python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-postproc --no-eager-line "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 -c "import json,sys; d=json.load(open('gpurun_out/ab_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config'].get('stream_layout'), d.get('latency_ms_per_batch'))"; }
MSPI_LIB_PATH=$PWD/tools/scratch/libmspi_hip_base.so run base1
run new1
MSPI_LIB_PATH=$PWD/tools/scratch/libmspi_hip_base.so run base2
run new2
