#!/usr/bin/env python3
"""profiles/pmc_sq.sh output (gpurun_out/<round>_sq_counters.txt) → the committed profiles/<round>_sq_counters.txt: a header, the raw
per-launch averages with the kernel names tidied, and the derived fractions per kernel.

    python profiles/sq_derive.py r5"""
import ast
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r5"
src = os.path.join(ROOT, "gpurun_out", f"{rnd}_sq_counters.txt")
rows, out = {}, []
for line in open(src):
    line = line.rstrip("\n")
    m = re.match(r"^(p[12])\s+[a-z: ]*?(k_\S.*?)\s+(\{.*\})\s+launches\s+(\d+)$", line)   # (the shell script cuts the name to 34 characters: "oid lde::k_…")
    if line.startswith("##"):
        out.append(line)
    elif m:
        p, name, d, n = m.group(1), m.group(2), ast.literal_eval(m.group(3)), int(m.group(4))
        out.append(f"{p} {name} {d} launches {n}")
        rows.setdefault(name, {}).update(d)
hdr = f"""# SQ counter passes (profiles/pmc_sq.sh via abl/collect_{rnd}.sh; per-launch averages; PMC only, no tracing) — round {rnd[1:]}.
# Template arguments: k_mlpb<SOLVER, D', ACT, DISC, ADJ>, k_mlpc<SOLVER, ACT, DISC, ADJ>: <…, false, false> the forward solve, <…, false, true> the
# continuous adjoint's reverse-time solve, <…, true, true> the discrete sweep (LDE_SENSE_DISCRETE); k_mlp64 / k_mlp64_adj / k_mlp64_disc likewise.
# (rocprofv3 7.2 crashes at process exit behind cooperative launches — the coupled workloads — after it has written its CSVs.)
# The last two sections: the metric's kernels — forward k_pend_forward_lp<REC> (round 6: lane pairs, Nyström form, four dense-output waves), pullbacks
# k_pend_adjoint_disc_tp (discrete, the default: steps side by side) and k_pend_adjoint_fused (continuous, time-parallel)."""
der = ["#", "# Derived (per launch): issue fraction = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; LDS wait = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES; "
       "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4·SQ_BUSY_CYCLES)"]
for name, d in rows.items():
    if "SQ_WAVE_CYCLES" not in d or "SQ_INSTS_VALU" not in d:
        continue
    wc = d["SQ_WAVE_CYCLES"]
    der.append(f"#   {name:34s} issue {d['SQ_ACTIVE_INST_ANY'] / wc:.2f}  LDS wait {d['SQ_WAIT_INST_LDS'] / wc:.2f}  "
               f"MFMA busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * max(d['SQ_BUSY_CYCLES'], 1)):.2f}  VALU {d['SQ_INSTS_VALU'] / 1e6:7.1f} M  "
               f"LDS {d['SQ_INSTS_LDS'] / 1e6:6.1f} M  MFMA {d['SQ_INSTS_MFMA'] / 1e6:5.2f} M  SALU {d['SQ_INSTS_SALU'] / 1e6:6.1f} M  "
               f"bank conflicts / LDS active {d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1):.2f}")
open(os.path.join(ROOT, "profiles", f"{rnd}_sq_counters.txt"), "w").write("\n".join([hdr] + out + der) + "\n")
print("\n".join(der))
