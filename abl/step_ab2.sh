#!/bin/bash
# round-3 host-side savings of the captured GOKU step, alternating on ONE box: the encoder as one autograd node, the library's ε
# generator, the constant seed gradient / no materialised retcode gradient — against their switches
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_loss.py tests/test_gpu_rnn.py tests/test_gpu_chain.py tests/test_gpu_graph_step.py tests/test_gpu_mixed_step.py tests/test_gpu_training.py tests/test_gpu_api.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
run() { env "$@" python bench.py --workload goku_step --no-cpu-baseline --steps 300 --warmup 30 ${DT:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"; }
run A=0 >/dev/null
for rep in 1 2 3; do echo "f32 new $(run A=0)  torch-rng $(run LDE_NATIVE_RNG=0)  separate-encoder $(run LDE_ENCODER_FUSED=0)  both-off $(run LDE_NATIVE_RNG=0 LDE_ENCODER_FUSED=0)"; done
export DT="--dtype mixed"
for rep in 1 2 3; do echo "mixed new $(run A=0)  torch-rng $(run LDE_NATIVE_RNG=0)  separate-encoder $(run LDE_ENCODER_FUSED=0)  both-off $(run LDE_NATIVE_RNG=0 LDE_ENCODER_FUSED=0)"; done
