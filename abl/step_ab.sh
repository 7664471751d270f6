#!/bin/bash
# goku_step A/B on ONE box, back to back (box-to-box variation is ≈ 8 %): the batched weight refresh and the issue order of the
# encoder branches, each against its switch; a throw-away run first (a fresh box's first process is slow).
cd "$GRAFT_REPO_ROOT"
run() { env "$@" python bench.py --workload goku_step --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms_per_step'])"; }
run A=0 > /dev/null
for rep in 1 2 3; do
  echo "default:            $(run A=0)"
  echo "LDE_BENCH_REFRESH=0 $(run LDE_BENCH_REFRESH=0)"
  echo "LDE_STACKS_FIRST=0  $(run LDE_STACKS_FIRST=0)"
  echo "both off            $(run LDE_BENCH_REFRESH=0 LDE_STACKS_FIRST=0)"
done
