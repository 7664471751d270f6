#!/bin/bash
# kernel launches and kernel time per training step (goku_step), from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r2_goku_step; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload goku_step --steps 100 --warmup 10 ${1:-} > $OUT/bench_trace.json 2>$OUT/trace.err
f=$(ls -t $OUT/trace/*/*kernel_stats.csv | head -1)
python3 - $f <<'EOF'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step", round(tot/110/1e6,3), " launches per step", round(sum(int(r["Calls"]) for r in rows)/110,1))
for r in rows[:30]:
    print(r["Name"][:72].ljust(72), round(int(r["Calls"])/110,1), round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
EOF
