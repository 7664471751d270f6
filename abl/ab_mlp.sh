#!/bin/bash
# A/B of variant libraries on one box, MLP workloads: abl/ab_mlp.sh <rounds> "<workload args>" <tag> …  → ms per step, solve-kernel ms of forward and adjoint
rounds=$1; shift; wl="$1"; shift
for r in $(seq 1 $rounds); do
  for t in "$@"; do
    LDE_LIB_PATH=$PWD/abl/liblde_$t.so python3 bench.py $wl --steps 20 --warmup 5 --no-cpu-baseline --no-other-sensealg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernel_ms',{})
print('$t $wl  step %.4f ms  fwd %.4f  adj %.4f  %.3g traj/s  frac %.3f' % (d['ms_per_step'], k.get('lde_forward',0), k.get('lde_adjoint',0), d['value'], d['roofline']['frac']))"
  done
done
