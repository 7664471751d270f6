#!/bin/bash
# re-collect the evidence of the two goku_step workloads only (abl/collect_all.sh does everything)
R=${1:-r3}
cd "$GRAFT_REPO_ROOT"
python bench.py --workload goku_step --steps 30 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
bash profiles/collect.sh ${R}_goku_step --workload goku_step --steps 100 --warmup 10 > /dev/null 2>&1
python bench.py --workload goku_step --steps 100 --warmup 10 > gpurun_out/bench_goku_step.json 2>/dev/null
bash profiles/collect.sh ${R}_goku_step_mixed --workload goku_step --dtype mixed --steps 100 --warmup 10 > /dev/null 2>&1
python bench.py --workload goku_step --dtype mixed --steps 100 --warmup 10 > gpurun_out/bench_goku_step_mixed.json 2>/dev/null
tail -c 300 gpurun_out/bench_goku_step_mixed.json
