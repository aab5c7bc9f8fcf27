#!/usr/bin/env python3
"""gpurun_out/inst_census.txt (tools/inst_census.sh) -> the per-wave-ray table of profiles/rNN_inst_census.txt.
A counter row is `kernel  COUNTER  mean  sum  instances` (instances = shader engines x dispatches of the run); per wave-ray = sum / dispatches /
(rays per launch / 64).   python tools/inst_census_table.py gpurun_out/inst_census.txt [rays per launch, default 20e6]"""
import re
import sys

path = sys.argv[1]
rays = float(sys.argv[2]) if len(sys.argv) > 2 else 20e6
COLS = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64",
        "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR"]
rows, cur, calls = {}, None, {}
for line in open(path):
    m = re.match(r"== max_hits (\d+) view (\w+)", line)
    if m:
        cur = (m.group(2), int(m.group(1)))
        rows[cur] = {}
        continue
    if cur is None or "halo_trace_kernel" not in line:
        continue
    f = line.split()
    m = re.search(r"\)\s+(SQ_\w+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$", line)
    if m:
        rows[cur][m.group(1)] = float(m.group(3))
    else:
        m = re.search(r"\)\s+(\d+)\s+[\d.]+\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s+[\d.]+%", line)
        if m:
            calls[cur] = int(m.group(1))
            rows[cur]["avg_us"] = float(m.group(2))
per = {}
print("%-22s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s" % ("max_hits / view", "VALU", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT", "SALU", "LDS", "SMEM",
                                                                        "VMEM_WR", "other", "us/launch"))
for key in sorted(rows, key=lambda k: (k[0] != "normal", k[1])):
    r, n = rows[key], max(calls.get(key, 2), 1)
    v = {c: r.get(c, 0.0) / n / (rays / 64.0) for c in COLS}
    other = v["SQ_INSTS_VALU"] - sum(v[c] for c in COLS[1:8])
    per[key] = dict(v, other=other)
    print("%-22s %s %9.1f %9.1f" % ("%d / %s" % (key[1], key[0]), " ".join("%9.1f" % v[c] for c in COLS), other, r.get("avg_us", 0.0)))
if ("normal", 7) in per and ("normal", 4) in per and ("normal", 1) in per:
    a, b, one = per[("normal", 7)], per[("normal", 4)], per[("normal", 1)]
    it = {c: (a[c] - b[c]) / 3.0 for c in a}
    print("\nper interaction (rows 7 - 4, / 3):  VALU %.1f  ADD_F32 %.1f  MUL_F32 %.1f  FMA_F32 %.1f  TRANS_F32 %.1f  INT32 %.1f  INT64 %.1f  CVT %.1f  other %.1f  SALU %.1f" % (
        it["SQ_INSTS_VALU"], it["SQ_INSTS_VALU_ADD_F32"], it["SQ_INSTS_VALU_MUL_F32"], it["SQ_INSTS_VALU_FMA_F32"], it["SQ_INSTS_VALU_TRANS_F32"], it["SQ_INSTS_VALU_INT32"],
        it["SQ_INSTS_VALU_INT64"], it["SQ_INSTS_VALU_CVT"], it["other"], it["SQ_INSTS_SALU"]))
    fixed = one["SQ_INSTS_VALU"] - it["SQ_INSTS_VALU"]
    print("per-ray fixed part (row 1 - one interaction): VALU %.0f of %.0f at max_hits 7 (%.0f %%); seven interactions %.0f" % (
        fixed, a["SQ_INSTS_VALU"], 100.0 * fixed / a["SQ_INSTS_VALU"], 7 * it["SQ_INSTS_VALU"]))
