#!/bin/bash
# the kernel sequence of one captured goku_step (mixed): abl/step_timeline.py on a fresh kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_loss.py tests/test_gpu_rnn.py tests/test_gpu_chain.py tests/test_gpu_graph_step.py tests/test_gpu_mixed_step.py tests/test_gpu_training.py tests/test_gpu_api.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --workload goku_step --dtype mixed --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/tl/bench.json 2> gpurun_out/tl/err.txt
python abl/step_timeline.py gpurun_out/tl
