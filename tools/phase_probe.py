"""Where do the trace kernel's cycles go?  Runs workloads on a -DHALO_PROBE build of the library (per-wave shader-clock stamps
at phase boundaries, summed over all waves) and prints each phase's share of the waves' resident cycles.
  HALO_PROBE=1 python -m ice_halo_sim_amd.build && HALO_LIB=ice_halo_sim_amd/libhalo_hip_probe.so python tools/phase_probe.py [cfg1|ms|stoch]
Probe builds perturb the kernel (stamps cost ~40 cycles each and pin the scheduler); read the shares, not the absolute time."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HALO_LIB", os.path.join(ROOT, "ice_halo_sim_amd", "libhalo_hip_probe.so"))
from ice_halo_sim_amd import abi, scenes  # noqa: E402
from ice_halo_sim_amd.backend import HipTraceBackend, load_library  # noqa: E402
from tests._oracle_backend import run_session  # noqa: E402

NAMES = ["streams+wavelength", "orientation sample", "rotation matrix", "sun cone + R^T", "entry pick", "fresnel", "emit: rotate/filter/gate",
         "emit: project", "emit: accumulate", "slab search + advance", "kernel prologue/epilogue", "TOTAL (wave resident)", "pool: stage shape record", "binned: flush hit buffer", "prologue (zero cache, stage tables)", "final drain of the exit queue"]


def dump(reset=True):
    L = load_library()
    out = (C.c_ulonglong * 16)()
    assert L.halo_probe_dump(out, 1 if reset else 0) == 0
    return list(out)


def run(label, sc, rd, wl, n):
    hb = HipTraceBackend(device=0, seed=42)
    run_session(hb, sc, rd, wl, n)          # warm-up
    hb.sync()
    dump(True)
    st = run_session(hb, sc, rd, wl, n)
    hb.sync()
    v = dump(True)
    hb.close()
    tot = max(v[11], 1)
    print("%s: %d rays, kernels %.3f ms; wave-resident cycles per wave-ray (64 rays) %.0f" % (label, n, sum(s.kernel_ms for s in st), tot / max(1, (n + 63) // 64 if n < (1 << 21) else 1)))
    for k in list(range(11)) + [12, 13, 14, 15]:
        print("   %-28s %6.2f %%" % (NAMES[k], 100.0 * v[k] / tot))
    print("   %-28s %6.2f %%   (loop overhead, stamps, divergence waits)" % ("unattributed", 100.0 * (tot - sum(v[:11]) - v[12] - v[13] - v[14] - v[15]) / tot))


which = sys.argv[1:] or ["cfg1"]
if "tiny" in which:
    for lg in (15, 16, 17, 18):
        run("configs[1] 550 nm, 2^%d rays" % lg, scenes.config2_scene(), scenes.config2_render(), scenes.wl_discrete(550.0), 1 << lg)
if "small" in which:   # one pass per workgroup: what a session of the reference's default GPU dispatch (2^18 rays, server.cpp:151) spends where
    run("configs[1] 550 nm, 2^18 rays", scenes.config2_scene(), scenes.config2_render(), scenes.wl_discrete(550.0), 1 << 18)
    run("configs[1] 550 nm, 2^20 rays", scenes.config2_scene(), scenes.config2_render(), scenes.wl_discrete(550.0), 1 << 20)
if "cfg1" in which:
    run("configs[1] 550 nm", scenes.config2_scene(), scenes.config2_render(), scenes.wl_discrete(550.0), 20_000_000)
if "ms" in which:
    run("configs[2] 550 nm (both layers)", scenes.config3_scene(), scenes.config2_render(), scenes.wl_discrete(550.0), 10_000_000)
if "stochp" in which:   # the pyramid variant (bench.py --config 4p)
    g = {"type": "gauss", "mean": 1.0, "std": 0.15}
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    e = scenes.entry(scenes.pyramid_crystal(0.1, 1.2, 0.5, upper_miller=(2, 3), face_distance=[g] * 6), scenes.axis(zenith=full, azimuth=full, roll=full), 100.0, 5)
    run("stochastic pyramid D65/31", scenes.scene([(0.0, [e])], max_hits=8),
        scenes.render(abi.LENS_RECTANGULAR, 2048, 1024, el=0.0, visible=abi.VISIBLE_FULL), scenes.wl_illuminant("D65", 31), 20_000_000)
if "stoch" in which:
    run("bench_config_stoch D65/31", scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8),
        scenes.render(abi.LENS_RECTANGULAR, 2048, 1024, el=0.0, visible=abi.VISIBLE_FULL), scenes.wl_illuminant("D65", 31), 20_000_000)
