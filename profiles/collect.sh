#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run from the repo root via gpurun):
#   1. kernel trace + stats (per-kernel durations)
#   2. PMC pass FETCH_SIZE   (separate run: TCC slots do not fit both, MI355X_MICROARCH.md §rocprofv3 PMC slots)
#   3. PMC pass WRITE_SIZE
# Outputs go to gpurun_out/prof_$TAG/{trace,fetch,write}; summarise with profiles/summarize.py.
set -u
TAG=${1:-r1}; shift || true
ARGS=${@:---steps 200 --warmup 20 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
