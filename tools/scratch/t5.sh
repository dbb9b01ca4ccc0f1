set -e
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "se_gate or dwconv or x3d" > gpurun_out/t5.log 2>&1 || { tail -40 gpurun_out/t5.log; exit 1; }
tail -1 gpurun_out/t5.log
ns() { tag=$1; shift; env "$@" python3 tools/northstar_targets.py > gpurun_out/ns_$tag.json 2> gpurun_out/ns_$tag.err; python3 -c "
import json
d=json.load(open('gpurun_out/ns_$tag.json')); x=d['x3d_conv_path_batch8']; print('$tag', x['ms_per_batch'], x['frac_of_hbm_peak'], {k:v['frac_of_hbm_peak'] for k,v in x['batches_in_flight'].items()})
"; }
ns wlds0 MSPI_DW_WLDS=0
ns wlds1 X=1
ns tileall MSPI_DW_TILE_ALL=1
ns wlds0b MSPI_DW_WLDS=0
ns wlds1b X=1
