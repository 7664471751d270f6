#!/bin/bash
# HBM counter bytes of the large-batch kernels (B = 2^20, both sensealgs): separate FETCH_SIZE / WRITE_SIZE passes + a kernel trace, per launch
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for sa in discrete continuous; do
  O=gpurun_out/lb_$sa; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 abl/lb_run.py 1048576 $sa > $O/run.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 abl/lb_run.py 1048576 $sa >> $O/run.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 abl/lb_run.py 1048576 $sa >> $O/run.log 2>&1
  grep ran $O/run.log | tail -1
done
python3 - <<'P'
import csv, glob, collections
for sa in ("discrete", "continuous"):
    O = f"gpurun_out/lb_{sa}"
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{O}/trace/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void lde::k_pend"): dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cnt = {}
    for what in ("fetch", "write"):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{O}/{what}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].startswith("void lde::k_pend"): acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
        cnt[what] = acc
    for k, d in dur.items():
        f_ = cnt["fetch"].get(k, [0]); w_ = cnt["write"].get(k, [0])
        # FETCH_SIZE / WRITE_SIZE are reported in KiB (profiles/summarize.py; MI355X_MICROARCH.md §HBM); these kernels load ≤ 8 bytes per lane: no ×2
        print(f"{sa:10s} {k:72s} avg {sum(d)/len(d):8.1f} us  FETCH {sum(f_)/len(f_)*1024/1e6:8.1f} MB  WRITE {sum(w_)/len(w_)*1024/1e6:8.1f} MB  ({len(d)} launches)")
P
