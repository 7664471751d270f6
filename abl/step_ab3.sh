#!/bin/bash
# one switch of the captured GOKU step (here: LDE_RECON_MSE_FWD) against its default: tests, then goku_step alternating on ONE box
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_chain.py tests/test_gpu_loss.py tests/test_gpu_graph_step.py tests/test_gpu_mixed_step.py tests/test_gpu_training.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
run() { env "$@" python bench.py --workload goku_step --no-cpu-baseline --steps 300 --warmup 30 ${DT:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"; }
run A=0 >/dev/null
for rep in 1 2 3; do echo "f32 fused-sum $(run A=0)  separate $(run LDE_RECON_MSE_FWD=0)"; done
export DT="--dtype mixed"
for rep in 1 2 3; do echo "mixed fused-sum $(run A=0)  separate $(run LDE_RECON_MSE_FWD=0)"; done
