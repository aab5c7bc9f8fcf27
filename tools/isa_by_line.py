#!/usr/bin/env python3
"""Static instruction census of ONE trace-kernel instantiation by SOURCE LINE (build container: needs hipcc only, no GPU).

A whole library of trace kernels takes ten minutes to build; one instantiation compiles to assembly in three seconds.  This tool compiles
`halo_trace_kernel<...>` (default: the headline instantiation <0, 3, true, 4, 1, 0, true>) with the build's own flags plus
-gline-tables-only, attributes every VALU instruction to the innermost source line of its .loc, and prints per line: instructions,
of them v_mov, v_cndmask, packed (v_pk_*), with an SGPR / VCC source operand, and 64-bit integer forms.  PC sampling does not exist on the
GPU boxes (rocprofv3 --pc-sampling: unsupported); the dynamic census is by class only (tools/inst_census.sh).  This table is what found, at
the end of round 6, that the SLP vectoriser's packed fp32 operations, the compiler's atomic optimizer around hand-aggregated atomics and
three nested `&&` branches cost the headline kernel 8 %.

  python tools/isa_by_line.py [--inst "0, 3, true, 4, 1, 0, true"] [--flags "-fno-slp-vectorize ..."] [--min 8] [--lines 2090:2420]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ice_halo_sim_amd", "csrc")
DEFAULT_FLAGS = "-fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None"   # ice_halo_sim_amd/build.py's flags for the kernel TUs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inst", default="0, 3, true, 4, 1, 0, true", help="template arguments MODE, GEOM, MONO, ACC, LENS, VIS, NOGATE")
    ap.add_argument("--flags", default=DEFAULT_FLAGS, help="compiler flags on top of -O3 -munsafe-fp-atomics (\"\" = the compiler's defaults)")
    ap.add_argument("--min", type=int, default=8, help="print lines with at least this many VALU instructions")
    ap.add_argument("--lines", default="", help="a:b = only source lines of halo_trace.inl in this range (then --min does not apply)")
    ap.add_argument("--keep", default="", help="write the annotated assembly here")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        tu = os.path.join(tmp, "one.hip")
        with open(tu, "w") as f:
            f.write('#include "halo_trace.inl"\nnamespace halo {\ntemplate __global__ void halo_trace_kernel<%s>(DispatchParams);\n}\n' % args.inst)
        out = args.keep or os.path.join(tmp, "one.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-std=c++17", "-fPIC", "-gline-tables-only", "-Rpass-analysis=kernel-resource-usage",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "--cuda-device-only", "-S", tu, "-o", out] + args.flags.split()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr[-3000:])
        for line in r.stderr.splitlines():
            if re.search(r" (VGPRs|SGPRs Spill|VGPRs Spill|ScratchSize|Occupancy|LDS Size)", line):
                print("#", line.split("remark:")[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip())
        text = open(out).read().split("\n")
    files, cur = {}, None
    cols = ("valu", "mov", "cndmask", "packed", "sgpr_src", "int64")
    cnt = {c: collections.Counter() for c in cols}
    skip_sgpr = ("v_readlane", "v_writelane", "v_readfirstlane", "v_cndmask", "v_cmp", "v_mbcnt")
    for line in text:
        t = line.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', t)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not re.match(r"v_[a-z0-9_]+", t):
            continue
        op = t.split()[0]
        body = t.split(";")[0]
        cnt["valu"][cur] += 1
        if op.startswith("v_mov"):
            cnt["mov"][cur] += 1
        if op.startswith("v_cndmask"):
            cnt["cndmask"][cur] += 1
        if op.startswith("v_pk_"):
            cnt["packed"][cur] += 1
        if re.match(r"v_(mad_u64|mad_i64|lshl_add_u64|lshlrev_b64|lshrrev_b64|ashrrev_i64|add_co|addc_co)", op):
            cnt["int64"][cur] += 1
        if not op.startswith(skip_sgpr):
            srcs = body.split(None, 1)[1].split(",")[1:] if " " in body else []
            if any(re.search(r"\bs\d+\b|s\[\d+:\d+\]|\bvcc", x) for x in srcs):
                cnt["sgpr_src"][cur] += 1
    print("# total:", "  ".join("%s %d" % (c, sum(cnt[c].values())) for c in cols))
    inl = [k for k, v in files.items() if v.endswith("halo_trace.inl")]
    src = open(os.path.join(CSRC, "halo_trace.inl")).read().split("\n")
    lo, hi = (int(x) for x in args.lines.split(":")) if args.lines else (0, 10 ** 9)
    other = collections.Counter()
    print("# %5s %5s %4s %4s %4s %4s %4s | source" % ("line", "valu", "mov", "cnd", "pk", "sgpr", "i64"))
    for key in sorted(cnt["valu"]):
        if key is None or key[0] not in inl:
            other[files.get(key[0], "?") if key else "?"] += cnt["valu"][key]
            continue
        if not (lo <= key[1] <= hi) or (not args.lines and cnt["valu"][key] < args.min):
            continue
        print("  %5d %5d %4d %4d %4d %4d %4d | %s" % (key[1], cnt["valu"][key], cnt["mov"][key], cnt["cndmask"][key], cnt["packed"][key], cnt["sgpr_src"][key],
                                                     cnt["int64"][key], src[key[1] - 1].strip()[:120] if key[1] > 0 else "(no line: prologue, copies at joins)"))
    print("# outside halo_trace.inl:", dict(other))


if __name__ == "__main__":
    main()
