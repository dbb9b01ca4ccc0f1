set -e
mkdir -p gpurun_out
C=gpurun_out/tune_if.json
rm -f $C
python3 bench.py --steps 20 --warmup 5 --tune-cache $C --no-cpu-baseline --no-roofline --no-postproc --no-eager-line > gpurun_out/if2.json 2> gpurun_out/if2.err
python3 bench.py --steps 20 --warmup 5 --tune-cache $C --no-cpu-baseline --no-roofline --no-postproc --no-eager-line --inflight 3 > gpurun_out/if3.json 2> gpurun_out/if3.err
python3 bench.py --steps 20 --warmup 5 --tune-cache $C --no-cpu-baseline --no-roofline --no-postproc --no-eager-line --inflight 4 > gpurun_out/if4.json 2> gpurun_out/if4.err
python3 bench.py --steps 20 --warmup 5 --tune-cache $C --no-cpu-baseline --no-roofline --no-postproc --no-eager-line --inflight 2 > gpurun_out/if2b.json 2> gpurun_out/if2b.err
for f in if2 if3 if4 if2b; do python3 -c "import json,sys; d=json.load(open('gpurun_out/$f.json')); print('$f', d['value'], d['ms_per_step'], d['config'].get('stream_layout'))"; grep -i "layout" gpurun_out/$f.err | tail -5; done
