set -e
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "se_fold or se_gate or dwconv_pool" > gpurun_out/t4.log 2>&1 || { tail -40 gpurun_out/t4.log; exit 1; }
tail -1 gpurun_out/t4.log
MSPI_SE_FOLD=0 python3 tools/northstar_targets.py > gpurun_out/ns_nofold.json 2> gpurun_out/ns_nofold.err
python3 tools/northstar_targets.py > gpurun_out/ns_fold.json 2> gpurun_out/ns_fold.err
python3 -c "
import json
for t in ('nofold','fold'):
    d=json.load(open('gpurun_out/ns_%s.json'%t)); x=d['x3d_conv_path_batch8']; print(t, x['ms_per_batch'], x['frac_of_hbm_peak'], {k:v['frac_of_hbm_peak'] for k,v in x['batches_in_flight'].items()})
"
