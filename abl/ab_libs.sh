#!/bin/bash
# A/B of variant libraries on ONE box: abl/ab_libs.sh <rounds> <tag> <tag> …   (abl/liblde_<tag>.so, see abl/variant_lib.sh)
# prints, per tag and round: K = 200 and K = 20 trajectories/s (M) and the forward kernel's back-to-back µs
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for t in "$@"; do
    for K in 200 20; do
      LDE_LIB_PATH=$PWD/abl/liblde_$t.so python3 bench.py --steps $K --warmup 5 --no-cpu-baseline --no-other-sensealg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t K=$K  %.2f M  step %.2f us  fwd %.2f us  adj %.2f us' % (d['value']/1e6, d['ms_per_step']*1e3, d['kernel_ms']['lde_forward']*1e3, d['kernel_ms']['lde_adjoint']*1e3))"
    done
  done
done
