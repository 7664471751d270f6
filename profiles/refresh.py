#!/usr/bin/env python3
"""Copy the newest profiles/collect.sh results from gpurun_out/ into profiles/ (summary JSON, kernel stats CSV, bench line).

    python profiles/refresh.py [--round r2] c2 c3 c4 goku_decoder goku_pendulum_b256

gpurun merges every call's files into the local gpurun_out/, so only the newest run of each sub-directory is used."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def newest_only(d):
    for sub in ("trace", "fetch", "write"):
        infos = sorted(glob.glob(f"{d}/{sub}/*/*_agent_info.csv"), key=os.path.getmtime)
        for f in infos[:-1]:
            pid = os.path.basename(f).split("_")[0]
            for g in glob.glob(os.path.join(os.path.dirname(f), pid + "_*")):
                os.remove(g)


args = sys.argv[1:]
RND = "r1"
if args and args[0] == "--round":
    RND, args = args[1], args[2:]
for w in args:
    d = f"{OUT}/prof_{RND}_{w}"
    newest_only(d)
    subprocess.run([sys.executable, f"{ROOT}/profiles/summarize.py", d, f"{ROOT}/profiles/{RND}_{w}_summary.json"],
                   stdout=subprocess.DEVNULL, check=True)
    ks = sorted(glob.glob(f"{d}/trace/*/*kernel_stats.csv"), key=os.path.getmtime)[-1]
    shutil.copy(ks, f"{ROOT}/profiles/{RND}_{w}_kernel_stats.csv")
    b = f"{OUT}/bench_{w}.json" if not w.startswith("goku_pendulum") else (f"{OUT}/bench_metric_discrete.json" if "discrete" in w else f"{OUT}/bench_metric.json")
    if os.path.exists(b):
        dst = f"{ROOT}/profiles/{RND}_{w}_bench.json"
        shutil.copy(b, dst)
        # the line was printed on the GPU box next to the PREVIOUS summary: re-attach the traffic figures from the one just written
        sys.path.insert(0, ROOT)
        import bench
        lines = open(dst).read().strip().splitlines()
        d = json.loads(lines[-1])
        roof = d.get("roofline") or {}
        if roof.get("bound") == "hbm" and w.startswith("goku_pendulum"):
            bench.attach_traffic(roof, "goku_pendulum_discrete" if "discrete" in w else "goku_pendulum", d["config"]["batch_per_gpu"], mlp=False, full_batch=True, rounds=(RND,),
                                 launched=roof.get("launched_kernels"))
        elif w.replace("_discrete", "") in ("c2", "c3", "c4", "latentode_ref"):
            bench.attach_traffic(roof, w, d["config"]["batch_per_gpu"], mlp=True, full_batch=True, rounds=(RND,), launched=roof.get("launched_kernels"))
        open(dst, "w").write(json.dumps(d) + "\n")
    s = json.load(open(f"{ROOT}/profiles/{RND}_{w}_summary.json"))
    print(w, {k.split("<")[0]: round(v.get("avg_ns", 0) / 1e3, 1) for k, v in s["kernels"].items() if k.startswith("k_")})
