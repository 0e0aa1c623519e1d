#!/bin/bash
# Collects the round's measurement evidence on the GPU box (run through gpurun from the repo root):
#   1. the default bench line (software-pipelined, HIP graph) and the same command under rocprofv3 --kernel-trace --stats
#   2. rocprofv3 --kernel-trace --stats of the UNOVERLAPPED run (`--pipelined 0 --no-overlap --no-graph`: every kernel
#      alone on the device, the execution bench.py's per-kernel HIP-event numbers come from) + per-family table
#   3. two PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of the unoverlapped run + per-family HBM bytes
#   4. per-op timings, the step with the FP ops, the batch sweep, BASELINE config 5, the training steps of configs 3 / 4
# usage: bash profiles/collect.sh TAG [a|b|c]  -> gpurun_out/prof_TAG/*  (copy what should be judged into profiles/)
#        (three parts, each within one gpurun call's time limit: a = 1-3, b = per-op / FP / sweep / config 5, c = training steps)
set -eu
TAG=${1:-r03}
PART=${2:-abc}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
if [[ $PART == *a* ]]; then
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 bench.py --cpu-scenes 0 --extras "" --verify-scenes 0 > $OUT/bench_under_rocprof.json
cp $(find $OUT/stats_default -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_default.csv
echo "default stats pass done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_alone -- python3 bench.py --cpu-scenes 0 --extras "" --verify-scenes 0 --pipelined 0 --no-overlap --no-graph > $OUT/bench_alone_under_rocprof.json
cp $(find $OUT/stats_alone -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_alone.csv
python3 profiles/summarize.py trace $(find $OUT/stats_alone -name "*kernel_trace.csv" | head -1) $OUT/family_durations_alone.json
echo "unoverlapped stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-scenes 0 --extras "" --verify-scenes 0 --pipelined 0 --no-overlap --no-graph > $OUT/pmc_fetch.log
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --cpu-scenes 0 --extras "" --verify-scenes 0 --pipelined 0 --no-overlap --no-graph > $OUT/pmc_write.log
echo "write pass done"
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
BATCH=$(python3 -c "import json;print(json.load(open('$OUT/bench.json'))['config']['scenes_per_gpu'])")
python3 profiles/summarize.py pmc $F $W $BATCH $OUT/pmc_traffic.json
cp $F $OUT/pmc_fetch_size.csv; cp $W $OUT/pmc_write_size.csv
rm -rf $OUT/stats_default $OUT/stats_alone $OUT/pmc_fetch $OUT/pmc_write
fi
if [[ $PART == *b* ]]; then
python3 bench_ops.py > $OUT/bench_ops.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ops -- python3 bench_ops.py --reps 5 > $OUT/bench_ops_under_rocprof.jsonl
cp $(find $OUT/stats_ops -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_ops.csv
rm -rf $OUT/stats_ops
echo "per-op pass done"
python3 bench.py --with-fp --cpu-scenes 0 --extras "" > $OUT/bench_withfp.json
for B in 1 16 64 128 512; do python3 bench.py --cpu-scenes 0 --extras "" --batch $B --pipelined $([ $B -ge 64 ] && echo 1 || echo 0) >> $OUT/sweep.jsonl; done
python3 bench.py --cpu-scenes 0 --extras "" --batch 256 --pipelined 0 >> $OUT/sweep.jsonl
echo "sweep done"
# BASELINE config 5 (dense 65536-point scenes): the full line, and every kernel alone under the profiler
python3 bench.py --config 5 --steps 5 --warmup 2 > $OUT/bench_config5.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg5 -- python3 bench.py --config 5 --batch 64 --steps 3 --warmup 1 --cpu-scenes 0 --verify-scenes 0 --pipelined 0 --no-overlap --no-graph > $OUT/bench_config5_alone_under_rocprof.json
cp $(find $OUT/stats_cfg5 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_config5_alone.csv
python3 profiles/summarize.py trace $(find $OUT/stats_cfg5 -name "*kernel_trace.csv" | head -1) $OUT/family_durations_config5_alone.json
rm -rf $OUT/stats_cfg5
# HBM bytes of the config-5 kernels (two PMC passes, every kernel alone)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc5_fetch -- python3 bench.py --config 5 --batch 32 --steps 2 --warmup 1 --cpu-scenes 0 --verify-scenes 0 --pipelined 0 --no-overlap --no-graph > $OUT/pmc5_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc5_write -- python3 bench.py --config 5 --batch 32 --steps 2 --warmup 1 --cpu-scenes 0 --verify-scenes 0 --pipelined 0 --no-overlap --no-graph > $OUT/pmc5_write.log
python3 profiles/summarize.py pmc $(find $OUT/pmc5_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc5_write -name "*counter_collection.csv" | head -1) 32 $OUT/pmc_traffic_config5.json
rm -rf $OUT/pmc5_fetch $OUT/pmc5_write
echo "config 5 done"
fi
if [[ $PART == *c* ]]; then
# configs 3 / 4 (per-rank part): the training steps around the hot path, the point stream alone for comparison
python3 bench_step.py > $OUT/bench_step_points.json
python3 bench_step.py --image --rpn-only > $OUT/bench_step_config3.json
python3 bench_step.py --image --rpn-only --sampler stock > $OUT/bench_step_config3_stock_sampler.json
python3 bench_step.py --image > $OUT/bench_step_config4.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_step -- python3 bench_step.py --image --steps 5 > $OUT/bench_step_config4_under_rocprof.json
cp $(find $OUT/stats_step -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_step_config4.csv
rm -rf $OUT/stats_step
for B in 1 2; do python3 bench_step.py --infer --batch $B --steps 30 >> $OUT/bench_step_infer.jsonl; done
echo "training-step passes done"
fi
