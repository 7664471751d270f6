#!/bin/bash
# large-batch A/B of variant libraries on one box: abl/lb_ab.sh <rounds> <tag> …  → forward / pullback ms at B = 2^16, 2^20 (bench.py --sweep), discrete default
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for t in "$@"; do
    LDE_LIB_PATH=$PWD/abl/liblde_$t.so python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-sensealg --sweep ${LB_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['batch_sweep']
print('$t ' + '  '.join('B=%s fwd %.3f bwd %.3f ms' % (k, v['fwd_ms'], v['bwd_ms']) for k,v in s.items()))"
  done
done
