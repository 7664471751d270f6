#!/bin/bash
# runtime knobs of hipGraph launches against the captured goku_step (mixed), one box: does any of them remove the per-replay
# __amd_rocclr_copyBuffer + ≈ 8 µs gap, or the ≈ 4.6 µs floor of a dependent kernel node?
cd "$GRAFT_REPO_ROOT"
run() { env "$@" python bench.py --workload goku_step --dtype mixed --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"; }
run A=0 >/dev/null
for rep in 1 2; do
  echo "base $(run A=0)"
  echo "HIP_FORCE_DEV_KERNARG=0 $(run HIP_FORCE_DEV_KERNARG=0)"
  echo "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $(run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0)"
  echo "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 $(run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1)"
  echo "DEBUG_HIP_GRAPH_BATCH_SIZE=1 $(run DEBUG_HIP_GRAPH_BATCH_SIZE=1)"
  echo "DEBUG_HIP_GRAPH_BATCH_SIZE=64 $(run DEBUG_HIP_GRAPH_BATCH_SIZE=64)"
  echo "DEBUG_HIP_KERNARG_COPY_OPT=0 $(run DEBUG_HIP_KERNARG_COPY_OPT=0)"
  echo "DEBUG_CLR_BLIT_KERNARG_OPT=1 $(run DEBUG_CLR_BLIT_KERNARG_OPT=1)"
  echo "ROC_USE_FGS_KERNARG=0 $(run ROC_USE_FGS_KERNARG=0)"
done
