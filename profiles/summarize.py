#!/usr/bin/env python3
"""Summarise a profiles/collect.sh run into one small JSON (committed under profiles/, read by bench.py).

    python profiles/summarize.py gpurun_out/prof_<tag> profiles/<name>.json

Units / corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
counts 64 B per 128-B request for wide coalesced streaming reads (×2 to get bytes) — that calibration is for
16-B-per-lane streams; other access widths are uncalibrated, so both the raw and the ×2 figure are stored, and
`fetch_bytes` (what bench.py prices `roofline.traffic` with) applies the ×2 ONLY to the kernels declared below as
16-byte-per-lane streaming readers; for every other kernel it is the raw count (a lower bound; ×2 is the upper one).
"""
import csv
import glob
import json
import sys
from collections import defaultdict


# kernels whose global reads are 16 B per lane, coalesced, streaming (the calibration's access shape)
WIDE16 = ("k_mlp_dw", "k_reduce_tiles", "k_chain_forward", "k_chain_backward", "k_chain_dw_b", "k_loss_", "k_adamw", "k_refresh_many",
          "k_build_frags")


def kernel_key(name):
    name = name.replace("void ", "")
    return name.split("(")[0].replace("lde::", "")


def main(src, dst):
    out = {"source": src, "kernels": {}}
    for f in glob.glob(f"{src}/trace/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            k = kernel_key(r["Name"])
            out["kernels"].setdefault(k, {})
            out["kernels"][k].update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), min_ns=float(r["MinNs"]),
                                     max_ns=float(r["MaxNs"]), pct=float(r["Percentage"]))
    for which in ("fetch", "write"):
        acc, cnt = defaultdict(float), defaultdict(int)
        for f in glob.glob(f"{src}/{which}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = kernel_key(r["Kernel_Name"])
                acc[k] += float(r["Counter_Value"])
                cnt[k] += 1
        for k in acc:
            kib = acc[k] / cnt[k]
            d = out["kernels"].setdefault(k, {})
            if which == "fetch":
                d["FETCH_SIZE_KiB_per_launch"] = kib
                d["fetch_bytes_raw"] = kib * 1024
                d["fetch_bytes_x2_gfx950"] = 2 * kib * 1024
                d["fetch_16B_per_lane"] = k.startswith(WIDE16)
                d["fetch_bytes"] = (2 if k.startswith(WIDE16) else 1) * kib * 1024
            else:
                d["WRITE_SIZE_KiB_per_launch"] = kib
                d["write_bytes"] = kib * 1024
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
