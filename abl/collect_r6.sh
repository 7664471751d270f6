#!/bin/bash
# Round 6's evidence (run on the GPU box from the repo root; then `python profiles/refresh.py --round r6 …` here):
#   rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes + the plain bench line for the metric and for every MLP workload, each with BOTH
#   definitions of the gradient (the workload's reference default first: LDE_SENSE_DISCRETE on the GOKU path, the continuous adjoint for a
#   NeuralODE); SQ counter passes (profiles/pmc_sq.sh) for the solve kernels of both.
R=${1:-r6}
cd "$GRAFT_REPO_ROOT"
python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1   # a fresh box runs its first process slow
bash profiles/collect.sh ${R}_goku_pendulum_discrete_b256 --sensealg discrete --steps 200 --warmup 20 --no-cpu-baseline --no-other-sensealg > /dev/null 2>&1
bash profiles/collect.sh ${R}_goku_pendulum_b256 --sensealg continuous --steps 200 --warmup 20 --no-cpu-baseline --no-other-sensealg > /dev/null 2>&1
python bench.py --steps 200 --warmup 20 --sweep > gpurun_out/bench_metric_discrete.json 2> gpurun_out/bench_metric.err
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_metric_steps20.json 2>> gpurun_out/bench_metric.err
python bench.py --steps 200 --warmup 20 --sensealg continuous --sweep > gpurun_out/bench_metric.json 2>> gpurun_out/bench_metric.err
for w in c2 c3 c4 latentode_ref; do
  bash profiles/collect.sh ${R}_$w --workload $w --sensealg continuous --steps 20 --warmup 5 --no-cpu-baseline --no-other-sensealg > /dev/null 2>&1
  python bench.py --workload $w --sensealg continuous --steps 20 --warmup 5 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  bash profiles/collect.sh ${R}_${w}_discrete --workload $w --sensealg discrete --steps 20 --warmup 5 --no-cpu-baseline --no-other-sensealg > /dev/null 2>&1
  python bench.py --workload $w --sensealg discrete --steps 20 --warmup 5 > gpurun_out/bench_${w}_discrete.json 2> gpurun_out/bench_${w}_discrete.err
done
: > gpurun_out/${R}_sq_counters.txt
for w in c2 c3 c4 latentode_ref; do
  echo "## $w --sensealg continuous" >> gpurun_out/${R}_sq_counters.txt
  bash profiles/pmc_sq.sh $w mlp --sensealg continuous >> gpurun_out/${R}_sq_counters.txt 2>&1
  echo "## $w --sensealg discrete" >> gpurun_out/${R}_sq_counters.txt
  bash profiles/pmc_sq.sh $w mlp --sensealg discrete >> gpurun_out/${R}_sq_counters.txt 2>&1
done
echo "## goku_pendulum (metric) --sensealg discrete (the default)" >> gpurun_out/${R}_sq_counters.txt
bash profiles/pmc_sq.sh goku_pendulum k_pend --sensealg discrete >> gpurun_out/${R}_sq_counters.txt 2>&1
echo "## goku_pendulum --sensealg continuous" >> gpurun_out/${R}_sq_counters.txt
bash profiles/pmc_sq.sh goku_pendulum k_pend --sensealg continuous >> gpurun_out/${R}_sq_counters.txt 2>&1
for d in f32 mixed; do
  python bench.py --workload goku_step --dtype $d > gpurun_out/bench_goku_step_$d.json 2> gpurun_out/bench_goku_step_$d.err
  python bench.py --workload goku_decoder --dtype $d > gpurun_out/bench_goku_decoder_$d.json 2> gpurun_out/bench_goku_decoder_$d.err
done
ls gpurun_out/bench_*.json | wc -l
python abl/metric_floor.py > gpurun_out/${R}_metric_floor.txt 2>&1   # (abl/liblde_pprof.so: built beforehand by abl/variant_lib.sh pprof "-DLDE_PEND_PROF=1")
