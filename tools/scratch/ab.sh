# same-box A/B of the bench line: ENV_A vs ENV_B settings (e.g. A="MSPI_SE_FOLD=0")
set -e
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-postproc --no-eager-line > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 -c "import json,sys; d=json.load(open('gpurun_out/ab_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config'].get('stream_layout'), d.get('latency_ms_per_batch'))"; }
run base1 ${A:-X=1}
run new1 ${B:-X=1}
run base2 ${A:-X=1}
run new2 ${B:-X=1}
