#!/bin/bash
# the wide-input first layer of the bf16 chains staged through LDS (chain_gemm_b_gx): parity tests, then one replay's kernel sequence
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_chain_bf16.py tests/test_gpu_chain.py tests/test_gpu_mixed_step.py tests/test_gpu_graph_step.py tests/test_gpu_rnn.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
run() { env "$@" python bench.py --workload goku_step --dtype mixed --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"; }
run A=0 > /dev/null
for rep in 1 2 3; do echo "mixed $(run A=0)"; done
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --workload goku_step --dtype mixed --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/tl/bench.json 2> gpurun_out/tl/err.txt
python abl/step_timeline.py gpurun_out/tl | grep -E "k_chain_forward_b<|k_chain_backward_b<|step"
