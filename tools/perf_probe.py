import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session

def run(label, sc, rd, n=10_000_000, reps=3, wl=550.0, **opts):
    hb = HipTraceBackend(device=0, seed=42, **opts)
    ms = []
    for r in range(reps):
        st = run_session(hb, sc, rd, scenes.wl_discrete(wl), n)
        ms.append(sum(s.kernel_ms for s in st))
    hb.close()
    best = min(ms)
    print("%-46s kernel %8.3f ms  %8.1f M rays/s" % (label, best, n / best / 1e3), flush=True)
    return best

sc = scenes.config2_scene()
rd = scenes.config2_render()
which = (sys.argv[1:] or ["base"]) if __name__ == "__main__" else []
if "base" in which:
    run("config2 1920x1080 upper (default)", sc, rd)
    run("  no accumulation (aggregate=2)", sc, rd, aggregate=2)
    run("  aggregate=0 mono=1", sc, rd, aggregate=0)
    run("  aggregate=3 (cache only, misses dropped)", sc, rd, aggregate=3)
    run("  aggregate=1 mono=0", sc, rd, mono=0)
    run("  aggregate=0 mono=0 (old)", sc, rd, aggregate=0, mono=0)
    run("  visible full", sc, scenes.render(1, 1920, 1080, fov=180, el=30, visible=abi.VISIBLE_FULL))
    run("  512x256", sc, scenes.config2_render(512, 256))
    run("  blocks_per_cu=4", sc, rd, blocks_per_cu=4)
    run("  blocks_per_cu=16", sc, rd, blocks_per_cu=16)
    run("  n=50M", sc, rd, n=50_000_000, reps=2)
    sc_fs = scenes.scene([(0.0, [scenes.entry(scenes.prism_crystal(1.3), scenes.axis(zenith={"type": "uniform", "mean": 90, "std": 360}, azimuth={"type": "uniform", "mean": 0, "std": 360}))])], max_hits=7)
    run("  full-sphere orientation", sc_fs, rd)
    run("  full-sphere, no accumulation", sc_fs, rd, aggregate=2)
if "copies" in which:
    for c in (1, 2, 4, 8, 16, 32, 64):
        run("config2 mono_copies=%d" % c, sc, rd, mono_copies=c)
        run("  512x256 mono_copies=%d" % c, sc, scenes.config2_render(512, 256), mono_copies=c)
    run("config2 mono_copies=8, plain atomics", sc, rd, mono_copies=8, aggregate=0)
    run("config2 mono_copies=32, plain atomics", sc, rd, mono_copies=32, aggregate=0)
if "stochd65" in which:
    # examples/bench_config_stoch.json: stochastic prism, D65, rectangular 2048x1024 full sky, max_hits 8, 10 M rays
    import time
    sc_s = scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8)
    rd_s = scenes.render(7, 2048, 1024, el=0, visible=2)
    import json
    for opts in json.loads(os.environ.get("STOCH_OPTS", '[{}, {"lambda_planes": 1}, {"bin": 0}]')):
        hb = HipTraceBackend(device=0, seed=42, **opts)
        for n in (10_000_000, 50_000_000):
            best = 1e9
            for r in range(3):
                hb.sync(); t0 = time.perf_counter()
                st = run_session(hb, sc_s, rd_s, scenes.wl_illuminant("D65", 64), n)
                hb.sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
            print("bench_config_stoch shape %s n=%dM: wall %.2f ms (%.0f M rays/s), kernels %.2f ms" % (opts, n // 1_000_000, best, n / best / 1e3, sum(s.kernel_ms for s in st)), flush=True)
        hb.close()
if "filter" in which:
    col = scenes.column_crystal_entry()
    def with_f(fid):
        e = type(col).from_buffer_copy(bytes(col)); e.filter_id = fid; return e
    table = [scenes.simple_filter(scenes.filter_term("raypath", raypath=[3, 5]), "P"),
             scenes.simple_filter(scenes.filter_term("entry_exit", entry=3, exit=5, min_len=2, max_len=4), "PBD"),
             scenes.simple_filter(scenes.filter_term("direction", az=180, el=20, radii=2.0), "", "filter_out"),
             scenes.complex_filter([[scenes.filter_term("raypath", raypath=[1, 3, 2])], [scenes.filter_term("entry_exit", entry=1), scenes.filter_term("crystal", crystal_id=3)],
                                    [scenes.filter_term("raypath", raypath=[3, 1, 5, 7, 4])]], "PBD"),
             scenes.simple_filter(scenes.filter_term("crystal", crystal_id=99), "", "filter_out")]
    for fid, name in ((0, "no filter (MODE 0)"), (1, "raypath P"), (2, "entry_exit PBD"), (3, "direction out"), (4, "complex PBD"), (5, "crystal 99 out (all pass)")):
        hb = HipTraceBackend(device=0, seed=42)
        hb.set_filters(table)
        sc_f = scenes.scene([(0.0, [with_f(fid)])], max_hits=7)
        best = 1e9
        for r in range(3):
            st = run_session(hb, sc_f, rd, scenes.wl_discrete(550.0), 10_000_000)
            best = min(best, st[0].kernel_ms)
        print("filter %-22s kernel %.3f ms  %.0f M rays/s  exits/root %.3f" % (name, best, 1e4 / best, st[0].exit_count / 1e7), flush=True)
        hb.close()
if "light" in which:
    # the reference's published GPU scene `bench_light_single_ms` (doc/performance-testing.md:465,501): prism h=1.2, random
    # orientation, D65, max_hits 7, dual fisheye equal area, resolution sweep — 130.5 M rays/s at 512x256 on its CUDA backend (RTX 4060 Ti)
    import time
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    sc_l = scenes.scene([(0.0, [scenes.entry(scenes.prism_crystal(1.2), scenes.axis(zenith=full, azimuth=full, roll=full), 1.0, 1)])], max_hits=7)
    for (w, h) in ((256, 128), (512, 256), (1024, 512), (2048, 1024)):
        rd_l = scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, w, h, visible=abi.VISIBLE_FULL)
        hb = HipTraceBackend(device=0, seed=42, **{"async": 1})
        n, reps = 50_000_000, 4
        for r in range(2):
            hb.sync(); t0 = time.perf_counter()
            for k in range(reps):
                run_session(hb, sc_l, rd_l, scenes.wl_illuminant("D65", 64), n)
            hb.sync(); dt = time.perf_counter() - t0
        st = hb.collect_stats()
        print("bench_light_single_ms shape %dx%d: %.1f M rays/s wall (%d x %d M rays), kernels %.2f ms per session" % (w, h, reps * n / dt / 1e6, reps, n // 1_000_000, st.kernel_ms / (2 * reps)), flush=True)
        hb.close()
if "bin" in which:
    full = scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 2048, 1024, visible=abi.VISIBLE_FULL)
    sc_s = scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8)
    rd_s = scenes.render(7, 2048, 1024, el=0, visible=2)
    for b in (0, 1):
        run("config2 10M bin=%d" % b, sc, rd, bin=b)
        run("config2 50M bin=%d" % b, sc, rd, n=50_000_000, reps=2, bin=b)
        run("config2 dual-fisheye full sky 2048x1024 10M bin=%d" % b, sc, full, bin=b)
        run("config2 dual-fisheye full sky 2048x1024 50M bin=%d" % b, sc, full, n=50_000_000, reps=2, bin=b)
        run("stochastic prism rect full sky 4M bin=%d" % b, sc_s, rd_s, n=4_000_000, bin=b)
        run("stochastic prism rect full sky 16M bin=%d" % b, sc_s, rd_s, n=16_000_000, bin=b)
if "illum" in which:
    wl_d65 = scenes.wl_illuminant("D65", 64)
    def run_wl(label, n, **opts):
        hb = HipTraceBackend(device=0, seed=42, **opts)
        best = 1e9
        for r in range(3):
            import time
            hb.sync(); t0 = time.perf_counter()
            st = run_session(hb, sc, rd, wl_d65, n)
            hb.sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
        print("%-46s wall %8.3f ms  kernel %8.3f ms  %8.1f M rays/s (wall)" % (label, best, sum(s.kernel_ms for s in st), n / best / 1e3), flush=True)
        hb.close()
    for n in (1_000_000, 10_000_000, 50_000_000):
        run_wl("D65 n=%dM xyz planes" % (n // 1_000_000), n, lambda_planes=0)
        run_wl("D65 n=%dM lambda planes" % (n // 1_000_000), n, lambda_planes=1)
    run_wl("D65 n=10M mono=0 copies=1", 10_000_000, lambda_planes=0, mono_copies=1)
    run_wl("D65 n=50M lambda planes, binned (two-level)", 50_000_000, lambda_planes=1, bin=1)
    run_wl("D65 n=10M lambda planes, binned (two-level)", 10_000_000, lambda_planes=1, bin=1)
if "bpc" in which:
    for bpc in (4, 5, 8, 12, 24):
        run("config2 50M blocks_per_cu=%d" % bpc, sc, rd, n=50_000_000, reps=2, blocks_per_cu=bpc)
if "hits" in which:
    for mh in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12):
        sc_h = scenes.config2_scene()
        sc_h.max_hits = mh
        run("config2 max_hits=%d" % mh, sc_h, rd)
    for mh in (1, 4, 7):
        sc_h = scenes.config2_scene()
        sc_h.max_hits = mh
        run("config2 max_hits=%d no-accum" % mh, sc_h, rd, aggregate=2)
if "ms" in which:
    run("config3 multi-scatter 10M", scenes.config3_scene(), rd)
if "stoch" in which:
    import time
    for hs in (0, 1):
        hb = HipTraceBackend(device=0, seed=42, host_shapes=hs)
        sc_s = scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8)
        rd_s = scenes.render(7, 2048, 1024, el=0, visible=2)
        run_session(hb, sc_s, rd_s, scenes.wl_discrete(550.0), 1_000_000)
        hb.sync(); t0 = time.perf_counter()
        st = run_session(hb, sc_s, rd_s, scenes.wl_discrete(550.0), 16_000_000)
        hb.sync(); dt = time.perf_counter() - t0
        print("stochastic prism 16M host_shapes=%d: wall %.1f ms (%.0f M rays/s), trace kernels %.2f ms" % (hs, dt * 1e3, 16 / dt, sum(s.kernel_ms for s in st)), flush=True)
        hb.close()
    run("stochastic prism 4M", scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8), scenes.render(7, 2048, 1024, el=0, visible=2), n=4_000_000)
if "pyr" in which:
    # stochastic pyramid (example crystal 5's Miller faces, gaussian face distances), full-sphere axis, rectangular full sky
    import time
    g = {"type": "gauss", "mean": 1.0, "std": 0.1}
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    pyr = scenes.entry(scenes.pyramid_crystal(0.1, 1.2, 0.5, upper_miller=(2, 3), face_distance=[g] * 6), scenes.axis(zenith=full, azimuth=full, roll=full), 1.0, 5)
    sc_p = scenes.scene([(0.0, [pyr])], max_hits=8)
    rd_p = scenes.render(7, 2048, 1024, el=0, visible=2)
    for opts in ({}, {"bin": 0}, {"aggregate": 2}):
        hb = HipTraceBackend(device=0, seed=42, **opts)
        for n in (4_000_000, 16_000_000):
            best = 1e9
            for r in range(3):
                hb.sync(); t0 = time.perf_counter()
                st = run_session(hb, sc_p, rd_p, scenes.wl_discrete(550.0), n)
                hb.sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
            print("stochastic pyramid %s n=%dM: wall %.2f ms (%.0f M rays/s), kernels %.2f ms, exits/root %.2f" % (opts, n // 1_000_000, best, n / best / 1e3, sum(s.kernel_ms for s in st), st[0].exit_count / n), flush=True)
        hb.close()
