#!/bin/bash
# evidence for the MLP configs only (c2, c3, c4): rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes and the bench line
cd "$GRAFT_REPO_ROOT"
for w in c2 c3 c4; do
  bash profiles/collect.sh r2_$w --workload $w --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
ls gpurun_out/bench_*.json | wc -l
