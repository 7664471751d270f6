#!/bin/bash
# goku_step A/B on ONE box, alternating (run-to-run variation on a box is ±5 %, box to box ≈ 8 %): this session's host-side
# changes (one weight-refresh launch per step, fused sample+KL / loss additions, encoder stacks issued first, one-launch ADAMW) against their
# switches; a throw-away run first (a fresh box's first process is slow). Prints per-run ms and the medians.
cd "$GRAFT_REPO_ROOT"
run() { env "$@" python bench.py --workload goku_step --no-cpu-baseline --steps 300 --warmup 30 ${DT:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])"; }
run A=0 > /dev/null
new=(); old=()
for rep in 1 2 3 4 5 6; do
  n=$(run A=0); o=$(run LDE_BENCH_REFRESH=0 LDE_STACKS_FIRST=0 LDE_FUSED_LOSS=0 LDE_NATIVE_ADAM=0)
  echo "new $n   old $o"
  new+=($n); old+=($o)
done
python - "${new[@]}" -- "${old[@]}" <<'PY'
import sys, statistics as st
a = sys.argv[1:]; i = a.index("--")
n, o = list(map(float, a[:i])), list(map(float, a[i+1:]))
print("median new %.4f ms, old %.4f ms (min %.4f / %.4f)" % (st.median(n), st.median(o), min(n), min(o)))
PY
