set -e
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "dwconv or x3d" > gpurun_out/t11.log 2>&1 || { tail -40 gpurun_out/t11.log; exit 1; }
MSPI_DW_TILE_RB=2 timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "dwconv" > gpurun_out/t11b.log 2>&1 || { tail -40 gpurun_out/t11b.log; exit 1; }
tail -1 gpurun_out/t11.log; tail -1 gpurun_out/t11b.log
ns() { tag=$1; shift; env "$@" python3 tools/northstar_targets.py > gpurun_out/ns_$tag.json 2> gpurun_out/ns_$tag.err; python3 -c "
import json
d=json.load(open('gpurun_out/ns_$tag.json')); x=d['x3d_conv_path_batch8']; print('$tag', x['ms_per_batch'], x['frac_of_hbm_peak'], {k:v['frac_of_hbm_peak'] for k,v in x['batches_in_flight'].items()})
"; }
ns rb1 MSPI_DW_TILE_RB=1
ns auto X=1
ns rb2 MSPI_DW_TILE_RB=2
ns rb1b MSPI_DW_TILE_RB=1
ns autob X=1
