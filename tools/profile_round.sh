#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel trace of the same command, HBM counter passes.
# usage: tools/profile_round.sh <tag>      (writes gpurun_out/prof_<tag>/...)
set -e
TAG=${1:-r03}
PART=${2:-all}      # a: bench + traces + HBM / MFMA counters;  b: mvitv2s MFMA, LDS conflicts, north-star targets, summaries (needs a's output)
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CACHE=$OUT/tune_x3dl.json
if [ "$PART" != b ]; then
rm -f $CACHE
python3 bench.py --steps 20 --warmup 5 --tune-cache $CACHE > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"; cat $OUT/bench.json
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r --output-format csv -- python3 bench.py --steps 20 --warmup 5 --tune-cache $CACHE --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
echo "trace done"
export MSPI_STREAMS=0
rocprofv3 --kernel-trace --stats -d $OUT/trace_serial -o r --output-format csv -- python3 bench.py --steps 20 --warmup 5 --inflight 1 --tune-cache $CACHE --no-cpu-baseline > $OUT/bench_under_rocprof_serial.json 2> $OUT/trace_serial.err
echo "serial trace done"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-graph --tune-cache $CACHE --no-cpu-baseline --no-roofline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o r --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-graph --tune-cache $CACHE --no-cpu-baseline --no-roofline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o r --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-graph --tune-cache $CACHE --no-cpu-baseline --no-roofline > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err
echo "mfma done"
python3 profiles/summarize.py $OUT $TAG > $OUT/summary.log 2>&1 || true
python3 tools/mfma_busy_summary.py $OUT/pmc_mfma "x3dl+audio B=8 (bench.py --no-graph)" > profiles/${TAG}_mfma_busy.csv || true
cp $OUT/bench.json profiles/${TAG}_bench.json; cp $OUT/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
cp $OUT/bench_under_rocprof_serial.json profiles/${TAG}_bench_under_rocprof_streams0.json
for d in trace trace_serial; do f=$(find $OUT/$d -name "*_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/${TAG}_rocprofv3_kernel_stats$([ $d = trace_serial ] && echo _streams0).csv; done
fi
if [ "$PART" != a ]; then
# the north star's second kernel target is measured on ITS workload: MViTv2-S attention (appended to part a's table: copy
# gpurun_out/profiles_$TAG/* into profiles/ between the two calls when they run on different boxes)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma_mvit -o r --output-format csv -- python3 bench.py --model mvitv2s --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $OUT/pmc_mfma_mvit.json 2> $OUT/pmc_mfma_mvit.err
echo "mfma (mvitv2s) done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_lds -o r --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err
echo "lds done"
python3 tools/mfma_busy_summary.py $OUT/pmc_mfma_mvit "mvitv2s+audio B=8 (bench.py --model mvitv2s --no-graph)" | tail -n +2 >> profiles/${TAG}_mfma_busy.csv || true
python3 tools/lds_conflict_summary.py $OUT/pmc_lds "x3dl+audio B=8 (bench.py --no-graph)" > profiles/${TAG}_lds_conflicts.csv || true
python3 tools/northstar_targets.py > profiles/${TAG}_northstar.json 2> $OUT/northstar.err || true
fi
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
# keep only the small summaries (the raw traces are tens of MB)
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
