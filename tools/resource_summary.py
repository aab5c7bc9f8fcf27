"""Per-kernel resource table from the hipcc remarks kept by ice_halo_sim_amd/build.py (build/resource_usage_*.txt).
usage: python tools/resource_summary.py [build_dir] [name filter]"""
import os
import re
import subprocess
import sys

bdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "ice_halo_sim_amd", "build")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for fn in sorted(os.listdir(bdir)):
    if not fn.startswith("resource_usage_"):
        continue
    cur = None
    for line in open(os.path.join(bdir, fn)):
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
except Exception:
    dem = names
print("%-5s %-5s %-7s %-4s %-7s %s" % ("VGPR", "AGPR", "scratch", "occ", "LDS", "kernel"))
for r, d in zip(rows, dem):
    d = re.sub(r"^void halo::", "", d).replace("(halo::DispatchParams)", "")
    if flt and flt not in d:
        continue
    print("%-5s %-5s %-7s %-4s %-7s %s" % (r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                                          r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?"), d))
