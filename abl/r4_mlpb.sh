#!/bin/bash
# round 4: k_mlpb (lde_mlpb.h) — parity subset, bench lines and the in-kernel phase profile of c2 / latentode_ref in one gpurun call
# usage: abl/r4_mlpb.sh <tag>
tag=${1:-x}
timeout 900 python -m pytest tests/test_gpu_mlp.py -x -q -k "families and (c2_rk4 or d12_h150)" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_golden.py -x -q -k "c2_latentode or latentode_ref" 2>&1 | tail -4
for w in c2 latentode_ref; do timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r4_${tag}_$w.json; done
python - <<PY
import json
for w in ("c2","latentode_ref"):
    d=json.load(open("gpurun_out/r4_${tag}_%s.json"%w)); r=d["roofline"]; print(w, round(d["ms_per_step"],4), "adj solve", round(r["avg_launch_ms"],4), "tail", round(r.get("tail",{}).get("avg_launch_ms",0),4), d.get("solver_stats",{}).get("adjoint"))
PY
if [ -f latentdiffeq.jl_amd/liblde_prof.so ]; then
  for w in c2 latentode_ref; do LDE_LIB_PATH=$PWD/latentdiffeq.jl_amd/liblde_prof.so timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -a "prof" | tail -2; done
fi
