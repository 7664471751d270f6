#!/bin/bash
# a subset of abl/collect_all.sh:   abl/collect_some.sh <round tag> <workload> [<workload> …]     (metric | c2 | c3 | c4 | latentode_ref | goku_step | goku_step_mixed)
R=$1; shift
cd "$GRAFT_REPO_ROOT"
python bench.py --workload goku_step --steps 30 --warmup 10 --no-cpu-baseline > /dev/null 2>&1   # a fresh box runs its first process ≈ 8 % slow
for w in "$@"; do
  if [ "$w" = metric ]; then
    bash profiles/collect.sh ${R}_goku_pendulum_b256 --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
    python bench.py --steps 200 --warmup 20 --sweep > gpurun_out/bench_metric.json 2> gpurun_out/bench_metric.err
  elif [ "$w" = goku_step ] || [ "$w" = goku_step_mixed ]; then
    dt=""; [ "$w" = goku_step_mixed ] && dt="--dtype mixed"
    bash profiles/collect.sh ${R}_$w --workload goku_step $dt --steps 100 --warmup 10 > /dev/null 2>&1
    python bench.py --workload goku_step $dt --steps 100 --warmup 10 > gpurun_out/bench_$w.json 2>/dev/null
  else
    bash profiles/collect.sh ${R}_$w --workload $w --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
    python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  fi
done
ls gpurun_out/bench_*.json | wc -l
