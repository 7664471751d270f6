#!/bin/bash
# the round's evidence: rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes and the plain bench line for every workload
#   abl/collect_all.sh [round tag, default r4]      (run on the GPU box from the repo root; then profiles/refresh.py --round <tag> …)
R=${1:-r4}
cd "$GRAFT_REPO_ROOT"
python bench.py --workload goku_step --steps 30 --warmup 10 --no-cpu-baseline > /dev/null 2>&1   # a fresh box runs its first process ≈ 8 % slow
bash profiles/collect.sh ${R}_goku_pendulum_b256 --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
python bench.py --steps 200 --warmup 20 --sweep > gpurun_out/bench_metric.json 2> gpurun_out/bench_metric.err
for w in c2 c3 c4 latentode_ref; do
  bash profiles/collect.sh ${R}_$w --workload $w --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
bash profiles/collect.sh ${R}_goku_decoder --workload goku_decoder --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
python bench.py --workload goku_decoder --steps 50 --warmup 10 > gpurun_out/bench_goku_decoder.json 2> gpurun_out/bench_goku_decoder.err
bash profiles/collect.sh ${R}_goku_decoder_mixed --workload goku_decoder --dtype mixed --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
python bench.py --workload goku_decoder --dtype mixed --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/bench_goku_decoder_mixed.json 2>/dev/null
bash profiles/collect.sh ${R}_goku_step --workload goku_step --steps 100 --warmup 10 > /dev/null 2>&1
python bench.py --workload goku_step --steps 100 --warmup 10 > gpurun_out/bench_goku_step.json 2>/dev/null
bash profiles/collect.sh ${R}_goku_step_mixed --workload goku_step --dtype mixed --steps 100 --warmup 10 > /dev/null 2>&1
python bench.py --workload goku_step --dtype mixed --steps 100 --warmup 10 > gpurun_out/bench_goku_step_mixed.json 2>/dev/null
python bench.py --workload c4 --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c4_b4096.json 2>/dev/null
python bench.py --workload c2 --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c2_b4096.json 2>/dev/null
ls gpurun_out/bench_*.json | wc -l
