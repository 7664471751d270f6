#!/bin/bash
# round 4: k_mlpc (lde_mlpc.h) — parity subset and bench line of c4 in one gpurun call.   usage: abl/r4_mlpc.sh <tag>
tag=${1:-x}
timeout 900 python -m pytest tests/test_gpu_mlp.py -x -q -k "families and (c4_coupled or d32_h128)" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_golden.py -x -q -k "c4_latentode" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k "c4" 2>&1 | tail -6
timeout 300 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r4_${tag}_c4.json
python - <<PY
import json
d=json.load(open("gpurun_out/r4_${tag}_c4.json")); r=d["roofline"]; print("c4", round(d["ms_per_step"],4), "adj solve", round(r["avg_launch_ms"],4), "tail", round(r.get("tail",{}).get("avg_launch_ms",0),4), d["kernel_ms"], d.get("solver_stats"))
PY
if [ -f latentdiffeq.jl_amd/liblde_prof.so ]; then
  LDE_LIB_PATH=$PWD/latentdiffeq.jl_amd/liblde_prof.so timeout 300 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -a "prof" | tail -2
fi
