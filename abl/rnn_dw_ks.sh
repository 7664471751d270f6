#!/bin/bash
# the recurrent stacks' weight-gradient tail in the captured step: merged initial-state sums (LDE_RNN_MERGE_S0) and the K-split of the
# cells' products (LDE_RNN_DW_KS; default min(8, 512 / (tiles × jobs))) — tests, then goku_step (mixed) per setting on ONE box
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_rnn.py tests/test_gpu_graph_step.py tests/test_gpu_mixed_step.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
run() { env "$@" python bench.py --workload goku_step --dtype mixed --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
run A=0 > /dev/null
for rep in 1 2; do
  echo "default $(run A=0)   no-merge $(run LDE_RNN_MERGE_S0=0)   ks1 $(run LDE_RNN_DW_KS=1)   ks2 $(run LDE_RNN_DW_KS=2)   ks4 $(run LDE_RNN_DW_KS=4)   ks8 $(run LDE_RNN_DW_KS=8)"
done
