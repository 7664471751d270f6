#!/bin/bash
# Round 5's evidence (run on the GPU box from the repo root; then `python profiles/refresh.py --round r5 …` here):
#   rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes + the plain bench line for the metric and for every MLP workload with
#   the continuous adjoint AND with LDE_SENSE_DISCRETE; SQ counter passes (profiles/pmc_sq.sh) for the MLP solve kernels of both.
R=${1:-r5}
cd "$GRAFT_REPO_ROOT"
python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1   # a fresh box runs its first process slow
bash profiles/collect.sh ${R}_goku_pendulum_b256 --steps 200 --warmup 20 --no-cpu-baseline --no-other-sensealg > /dev/null 2>&1
bash profiles/collect.sh ${R}_goku_pendulum_discrete_b256 --sensealg discrete --steps 200 --warmup 20 --no-cpu-baseline --no-other-sensealg > /dev/null 2>&1
python bench.py --steps 200 --warmup 20 --sweep > gpurun_out/bench_metric.json 2> gpurun_out/bench_metric.err
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_metric_steps20.json 2>> gpurun_out/bench_metric.err
python bench.py --steps 200 --warmup 20 --sensealg discrete --sweep > gpurun_out/bench_metric_discrete.json 2>> gpurun_out/bench_metric.err
for w in c2 c3 c4 latentode_ref; do
  bash profiles/collect.sh ${R}_$w --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-other-sensealg > /dev/null 2>&1
  python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  bash profiles/collect.sh ${R}_${w}_discrete --workload $w --sensealg discrete --steps 20 --warmup 5 --no-cpu-baseline --no-other-sensealg > /dev/null 2>&1
  python bench.py --workload $w --sensealg discrete --steps 20 --warmup 5 > gpurun_out/bench_${w}_discrete.json 2> gpurun_out/bench_${w}_discrete.err
done
: > gpurun_out/r5_sq_counters.txt
for w in c2 c3 c4 latentode_ref; do
  echo "## $w (continuous adjoint)" >> gpurun_out/r5_sq_counters.txt
  bash profiles/pmc_sq.sh $w mlp >> gpurun_out/r5_sq_counters.txt 2>&1
  echo "## $w --sensealg discrete" >> gpurun_out/r5_sq_counters.txt
  bash profiles/pmc_sq.sh $w mlp --sensealg discrete >> gpurun_out/r5_sq_counters.txt 2>&1
done
echo "## goku_pendulum (metric, continuous adjoint)" >> gpurun_out/r5_sq_counters.txt
bash profiles/pmc_sq.sh goku_pendulum k_pend >> gpurun_out/r5_sq_counters.txt 2>&1
echo "## goku_pendulum --sensealg discrete" >> gpurun_out/r5_sq_counters.txt
bash profiles/pmc_sq.sh goku_pendulum k_pend --sensealg discrete >> gpurun_out/r5_sq_counters.txt 2>&1
for d in f32 mixed; do
  python bench.py --workload goku_step --dtype $d > gpurun_out/bench_goku_step_$d.json 2> gpurun_out/bench_goku_step_$d.err
  python bench.py --workload goku_decoder --dtype $d > gpurun_out/bench_goku_decoder_$d.json 2> gpurun_out/bench_goku_decoder_$d.err
done
ls gpurun_out/bench_*.json | wc -l
