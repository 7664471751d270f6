#!/bin/bash
# HBM write bytes and SQ counters of the large-batch forward (B = 2^20, T = 50) per ring size: abl/pend_LB_pmc.sh "0 16 32"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for r in ${1:-0 16}; do
  export LDE_PEND_LB=$r
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_lb/w$r -- python3 abl/pend_LB_once.py > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d gpurun_out/pmc_lb/s$r -- python3 abl/pend_LB_once.py > /dev/null 2>&1
done
python3 - <<PY
import csv,glob
from collections import defaultdict
for d in sorted(glob.glob("gpurun_out/pmc_lb/*")):
    acc=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(lambda: defaultdict(int))
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][-44:]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
    for k,v in acc.items():
        if "pend_forward" in k: print(d[-4:],k,{a:round(b/cnt[k][a]) for a,b in v.items()})
PY
