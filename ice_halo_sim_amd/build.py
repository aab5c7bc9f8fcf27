"""Build libhalo_hip.so (gfx950 only) in-tree with hipcc.  `python -m ice_halo_sim_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
PROBE = bool(os.environ.get("HALO_PROBE"))   # phase-probe build (tools/phase_probe.py): its own file, never the shipped library
# HALO_BUILD_TAG=x builds libhalo_hip_x.so in build_x/ (A/B experiments: load it with HALO_LIB=...); HALO_DEFS="-DA=1 -DB" adds macros
TAG = "probe" if PROBE else os.environ.get("HALO_BUILD_TAG", "")
# HALO_BUILD_TAG=strict is a TESTED variant, not an experiment: the reference's roundings (separately rounded products and sums, IEEE division
# and square root in the Fresnel split) — libhalo_hip_strict.so, held to the unconditioned per-ray bars by tests/test_gpu_strict_variant.py
if TAG == "strict":
    os.environ["HALO_DEFS"] = (os.environ.get("HALO_DEFS", "") + " -DHALO_STRICT=1 -DHALO_FRESNEL=1").strip()
    os.environ["HALO_FP_CONTRACT"] = "off"
LIB = os.path.join(HERE, "libhalo_hip_%s.so" % TAG if TAG else "libhalo_hip.so")
SOURCES = ["halo_trace_m0.hip", "halo_trace_m1.hip", "halo_trace_m2.hip", "halo_trace_m3.hip", "halo_trace_m4.hip", "halo_kernels.hip", "halo_shapegen.hip", "halo_backend.cpp",
           "halo_host.cpp"]
NO_CONTRACT = {"halo_shapegen.hip"}   # geometry shared with the host: same rounding on both sides
HEADERS = ["halo_device.h", "halo_trace.inl", "halo_geom.h", "halo_host.hpp", "cie_tables.inc", os.path.join("..", "..", "include", "halo_trace.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cc = hipcc()
    bdir = os.path.join(HERE, "build_" + TAG if TAG else "build")
    os.makedirs(bdir, exist_ok=True)
    common = ["-std=c++17", "-fPIC", "-I", os.path.join(HERE, "..", "include")]

    def compile_one(src):
        obj = os.path.join(bdir, src + ".o")
        if src.endswith(".hip"):
            # -fno-slp-vectorize (round 6): the SLP vectoriser pairs independent fp32 operations into v_pk_add / v_pk_mul / v_pk_fma, which on gfx950
            # cost 4.5 cycles per wave64 instruction against 2 x 2.8 for the plain forms AND need their operands moved into register pairs first
            # (profiles/r06_micro_valu_rate_bench.txt; 95 packed instructions and 23 moves in the headline kernel): a net loss in an issue-bound
            # kernel.  Same values (a packed half rounds like the plain instruction); configs[1] 1.661 -> 1.588 ms per launch, configs[2] / [4] -3 / -2 %.
            # -amdgpu-atomic-optimizer-strategy=None: every same-address atomic of these kernels is wave-aggregated by hand already (ballot + mbcnt,
            # one lane adds: the hit log's cursor, the exit queue, the continuation shards) and all others have per-lane addresses; the compiler's
            # optimizer wraps the hand-aggregated ones in a second mbcnt / readfirstlane / multiply sequence (28 VALU + 38 SALU instructions in the
            # headline kernel, 3 SGPR spills).
            cmd = [cc, "--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-fno-slp-vectorize"]
            if not os.environ.get("HALO_ATOMIC_OPT"):   # (experiment knob: HALO_ATOMIC_OPT=1 leaves the compiler's atomic optimizer on, for the A/B)
                cmd += ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
            cmd += ["-Rpass-analysis=kernel-resource-usage"] + common
            if PROBE:
                cmd.append("-DHALO_PROBE=1")
            cmd += os.environ.get("HALO_DEFS", "").split()
            cmd += os.environ.get("HALO_HIPFLAGS", "").split()   # experiment knob: extra compiler flags for the kernel TUs
            for knob in ("HALO_MIN_WAVES", "HALO_MIN_WAVES_FILTER"):
                if os.environ.get(knob):
                    cmd.append("-D%s=%s" % (knob, os.environ[knob]))
            if src in NO_CONTRACT:
                cmd += ["-ffp-contract=off", "-fno-fast-math"]
            elif os.environ.get("HALO_FP_CONTRACT"):  # experiment knob: off | on | fast
                cmd.append("-ffp-contract=" + os.environ["HALO_FP_CONTRACT"])
        else:  # host tables must round like the reference's host build: no FMA contraction
            cmd = [cc, "-O2", "-ffp-contract=off", "-fno-fast-math", "-D__HIP_PLATFORM_AMD__"] + common + os.environ.get("HALO_DEFS", "").split()
        cmd += ["-c", os.path.join(CSRC, src), "-o", obj]
        # an object is rebuilt when it is missing, older than its source or a header it can see (the kernels' .inl only matters to the .hip
        # translation units), or was made by another command line (flags, macros)
        deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS if src.endswith(".hip") or not h.endswith(".inl")]
        stamp = obj + ".cmd"
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == " ".join(cmd) and \
                all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            return src, obj, subprocess.CompletedProcess(cmd, 0, "", "")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            with open(stamp, "w") as f:
                f.write(" ".join(cmd))
        if src.endswith(".hip") and r.returncode == 0:
            with open(os.path.join(bdir, "resource_usage_%s.txt" % src.split(".")[0]), "w") as f:
                f.write(r.stderr)
        return src, obj, r

    # the translation units are independent: compile them side by side (the three halo_trace_m*.hip dominate)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_one, SOURCES))
    objs = []
    for src, obj, r in results:
        if verbose or r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on " + src)
        objs.append(obj)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
