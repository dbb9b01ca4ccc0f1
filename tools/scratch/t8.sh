set -e
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-postproc --no-eager-line > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 -c "import json,sys; d=json.load(open('gpurun_out/ab_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config'].get('stream_layout'), d.get('latency_ms_per_batch'))"; }
run base1 X=1
run fuse1 MSPI_X3D_FUSE=1
run fuse0 MSPI_X3D_FUSE=0
run tileall MSPI_DW_TILE_ALL=1
run base2 X=1
run fuse1b MSPI_X3D_FUSE=1
run seam0 MSPI_X3D_SEAM=0
run base3 X=1
