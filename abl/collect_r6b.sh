#!/bin/bash
# Round 6, after the metric's forward kernel changed again (four dense-output waves): the metric's share of abl/collect_r6.sh only — kernel
# traces + FETCH / WRITE passes (both sensealgs), its bench lines, its SQ passes, the metric-floor stamps (prof build), and the whole-step lines
# (they launch the same kernel). The MLP workloads' evidence (profiles/r6_c*, r6_latentode_ref*) stands: those kernels did not change.
R=${1:-r6}
cd "$GRAFT_REPO_ROOT"
python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1   # a fresh box runs its first process slow
bash profiles/collect.sh ${R}_goku_pendulum_discrete_b256 --sensealg discrete --steps 200 --warmup 20 --no-cpu-baseline --no-other-sensealg > /dev/null 2>&1
bash profiles/collect.sh ${R}_goku_pendulum_b256 --sensealg continuous --steps 200 --warmup 20 --no-cpu-baseline --no-other-sensealg > /dev/null 2>&1
python bench.py --steps 200 --warmup 20 --sweep > gpurun_out/bench_metric_discrete.json 2> gpurun_out/bench_metric.err
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_metric_steps20.json 2>> gpurun_out/bench_metric.err
python bench.py --steps 200 --warmup 20 --sensealg continuous --sweep > gpurun_out/bench_metric.json 2>> gpurun_out/bench_metric.err
: > gpurun_out/${R}_sq_counters_metric.txt
echo "## goku_pendulum (metric) --sensealg discrete (the default)" >> gpurun_out/${R}_sq_counters_metric.txt
bash profiles/pmc_sq.sh goku_pendulum k_pend --sensealg discrete >> gpurun_out/${R}_sq_counters_metric.txt 2>&1
echo "## goku_pendulum --sensealg continuous" >> gpurun_out/${R}_sq_counters_metric.txt
bash profiles/pmc_sq.sh goku_pendulum k_pend --sensealg continuous >> gpurun_out/${R}_sq_counters_metric.txt 2>&1
for d in f32 mixed; do
  python bench.py --workload goku_step --dtype $d > gpurun_out/bench_goku_step_$d.json 2> gpurun_out/bench_goku_step_$d.err
done
python abl/metric_floor.py > gpurun_out/${R}_metric_floor.txt 2>&1
ls gpurun_out/bench_*.json | wc -l
