#!/bin/bash
# SQ counter passes for one bench workload (PMC only, no tracing): where do the wave-cycles go?
#   profiles/pmc_sq.sh <workload> [kernel-name filter] [extra bench args...]
set -u
W=${1:-c4}; shift || true
FILT=${1:-mlp}; shift || true
EXTRA=${@:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$W
rm -rf $OUT/p1 $OUT/p2   # (one workload's passes with and without --sensealg discrete share the directory: only this call's kernels are reported)
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-other-sensealg $EXTRA > $OUT/b1.json 2> $OUT/e1.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-other-sensealg $EXTRA > $OUT/b2.json 2> $OUT/e2.txt
python3 - <<PY
import csv,glob
from collections import defaultdict
for p in ("p1","p2"):
    acc=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(lambda: defaultdict(int))
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv"%p):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][-34:]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
    for k,v in acc.items():
        if "$FILT" in k: print(p,k,{a:round(b/cnt[k][a]) for a,b in v.items()}, "launches", max(cnt[k].values()))
PY
