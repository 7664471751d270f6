#!/bin/bash
# round 3, third part, as one A/B on ONE box: the captured GOKU step with every run-time switch of the part at its default against all of
# them off (one wave per stack, separate encoder nodes, torch's generator, separate reconstructor / loss nodes, separate sample launches,
# separate initial-state sums). Compile-time changes (staged first layer, hand-over, the update's own step count) stay on both sides.
cd "$GRAFT_REPO_ROOT"
OFF="LDE_RNN_PIPE=0 LDE_ENCODER_FUSED=0 LDE_NATIVE_RNG=0 LDE_RECON_MSE=0 LDE_SAMPLE_PAIR=0 LDE_RNN_MERGE_S0=0"
run() { env "$@" python bench.py --workload goku_step --no-cpu-baseline --steps 300 --warmup 30 ${DT:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"; }
run A=0 >/dev/null
for rep in 1 2 3; do echo "f32 on $(run A=0)  off $(run $OFF)"; done
export DT="--dtype mixed"
for rep in 1 2 3; do echo "mixed on $(run A=0)  off $(run $OFF)"; done
