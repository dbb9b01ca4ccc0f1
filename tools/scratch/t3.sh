set -e
mkdir -p gpurun_out
A="MSPI_SE_FOLD=0" bash tools/scratch/ab.sh
MSPI_SE_FOLD=0 python3 tools/northstar_targets.py > gpurun_out/ns_nofold.json 2> gpurun_out/ns_nofold.err
python3 tools/northstar_targets.py > gpurun_out/ns_fold.json 2> gpurun_out/ns_fold.err
python3 -c "
import json
for t in ('nofold','fold'):
    d=json.load(open('gpurun_out/ns_%s.json'%t)); x=d['x3d_conv_path_batch8']; print(t, x['ms_per_batch'], x['frac_of_hbm_peak'], {k:v['frac_of_hbm_peak'] for k,v in x['batches_in_flight'].items()})
"
