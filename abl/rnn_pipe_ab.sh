#!/bin/bash
# one wave per cell (LDE_RNN_PIPE=1, default) against the single wave per stack (=0): the recurrent tests, the isolated stack timings, then
# goku_step alternating on ONE box
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_rnn.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for p in 0 1; do echo "PIPE=$p"; LDE_RNN_PIPE=$p timeout 200 python abl/rnn_bench.py 2>&1 | tail -4; done
run() { env "$@" python bench.py --workload goku_step --no-cpu-baseline --steps 300 --warmup 30 ${DT:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"; }
run A=0 >/dev/null
for rep in 1 2 3; do echo "f32 pipe $(run LDE_RNN_PIPE=1)  single $(run LDE_RNN_PIPE=0)"; done
export DT="--dtype mixed"
for rep in 1 2 3; do echo "mixed pipe $(run LDE_RNN_PIPE=1)  single $(run LDE_RNN_PIPE=0)"; done
