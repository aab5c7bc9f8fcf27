#!/usr/bin/env python3
"""bench.py — rays/s of the MI355X trace hot path on BASELINE.json's workloads.

  python bench.py --gpus N --steps K --warmup W [--config 1|2|4|4d|4p|ref:<name>] [--repeats R] [--scaling weak|strong]
N > 1: the driver launches `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU, RCCL);
started WITHOUT a launcher (`python bench.py --gpus N`, WORLD_SIZE unset) the file launches those N ranks itself (self_launch).  One "step" = one pass of the hot path over one batch of the selected configuration:

  --config 1 (default, BASELINE `metric`)  configs[1]: single-scatter hex column (examples/config_example.json crystal 3),
             9 wavelengths x 50 M root rays, fisheye_equal_area 1920x1080 -> 450 M root rays per GPU per step
  --config 2  configs[2]: two-layer full multi-scattering (plate prob 1.0 over a randomly oriented column), 9 wavelengths x
             50 M roots per GPU per step (each root becomes ~4.7 second-layer rays)
  --config 4  configs[4] on the reference's file as shipped (examples/bench_config_stoch.json: stochastic PRISM, six
             gauss(1, 0.15) face distances, D65, rectangular 2048x1024 full sky, max_hits 8), D65 pool of 31 wavelengths,
             25 M rays per GPU per step (200 M over 8 GPUs)
  --config 4d the same file under the PRIMARY reading of "31 wavelengths" (SURVEY.md 8(d) item 5): a discrete 31-entry spectrum, 380..780 nm,
             weight = the D65 SPD at each wavelength (src/include/lumice.h:292-293 allows <= 255 entries) -> 31 sessions per step that share the
             25 M rays per GPU (per_wavelength_ray_num = ceil(25 M / 31) = 806 452 roots per session)
  --config 4p the pyramidal variant of the same (examples/config_example.json crystal 5 with the same face distances)
  --config ref:<name>  a config DOCUMENT of the reference, through the JSON reader (ice_halo_sim_amd.config), at its own resolution:
             ref:bench_light_single_ms, ref:ms_multi_crystal, ref:ms_multi_crystal_complex_filter, ref:ms_multi_crystal_filtered_bd —
             the reference's published GPU benchmark scenes (doc/performance-testing.md:465-468; fixture tests/golden/ref_e2e_configs.json),
             one step = 4 sessions x 50 M root rays (>= 200 M rays per timed step, the reference's steady-state rule) — and
             ref:config_example, examples/config_example.json as shipped (README quick start: 9 wavelengths x 50 M rays, renderer 4;
             fixture tests/golden/ref_example_configs.json)

With the default configuration on one GPU the JSON line also carries `other_configs`: short runs (3 warm-up steps, then 3 repeats of 3 timed
steps: median + CoV) of configs 2, 4, 4d, 4p and of the five reference documents — metric, ms/step, route and roofline of each — so that the
driver's record holds them; the headline fields are those of configs[1] alone.

Rays shard by index range (disjoint RNG counter ranges per rank, no data-path collective); every step ends with ONE RCCL
sum-reduce of the W*H*3(+4) fp32 accumulator to rank 0 (the reference's drain point, simulator.cpp:1409-1477).  Per-GPU work
is fixed -> "scaling": "weak" (the default); --scaling strong fixes the TOTAL instead (configs[3]'s "400 M rays sharded across 8"
read as a fixed job: rays per rank = rays / N).

Timing follows the reference's protocol (doc/performance-testing.md:109-131: >= 5 repeats, median + CoV): after W warm-up
steps the timed region — EXACTLY K steps between barrier + synchronize — is run R times (default 5); `value` and
`ms_per_step` are the MEDIAN repeat, `repeats` carries every repeat and the coefficient of variation.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel of the configuration against HBM (no dense contraction
-> no MFMA), as the contract asks: achieved = ALGORITHMIC bytes per launch (accumulator + continuation bytes, SURVEY 8(d)'s
fused definition) / mean launch duration from HIP events on the launch stream; peak 8 TB/s.  `bound` names the roof that actually BINDS the
kernel — `valu_issue` for every trace kernel of this repo (rays live in registers; `roofline.valu` prices the instruction stream with the
rates tools/valu_rate_bench.hip measures on the part) — and `hbm_frac` / `frac` keep the HBM fraction beside it.  `cpu_baseline` times the CPU oracle (a port of the reference's algorithm; the reference's own CPU path cannot be
built in this image without stand-ins for spdlog / nlohmann-json — DESIGN.md §5) on the host cores on a bounded sample of the
same workload — a reported baseline, not the target.
"""
import argparse
import json
import os
import statistics
import sys
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); the launcher
# exports it already — keep it if this file is started some other way
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_ROUND = "r06"  # committed rocprofv3 summaries this file reads counters from: profiles/<round>_bench<cfg>_*.txt

# The reference's legacy CPU path measured by the survey in the build container (compiled with shims, 6 worker
# threads, config_example-shaped scene): the only number that relates the oracle ("port") to the real reference.
SURVEY_REFERENCE_CPU = {"rays_per_s": 1.71e6, "threads": 6, "where": "build container (8 vCPU), SURVEY.md §6"}
# ... and the oracle timed in the same container on the same shape (tests/golden/CPU_CALIBRATION.md): 0.65 M rays/s on one thread against the
# reference's 0.312 M, 3.88 M on six against 1.71 M
ORACLE_OVER_REFERENCE = {"threads_1": 0.65 / 0.312, "threads_6": 3.88 / 1.71, "source": "tests/golden/CPU_CALIBRATION.md"}


def workload(cfg):
    """(scene, render, [wl sessions], rays per session, description, dominant kernel name) of a BASELINE configuration."""
    from ice_halo_sim_amd import abi, scenes
    if cfg == "1":
        return dict(scene=scenes.config2_scene(), render=scenes.config2_render(),
                    wls=[scenes.wl_discrete(w) for w in scenes.CONFIG_WAVELENGTHS_9], rays=50_000_000,
                    name="configs[1]: single-scatter hex column (prism h=1.3, zenith gauss(90,0.3)), 9 wavelengths x %d root rays per GPU per step, max_hits 7, fisheye_equal_area fov 180 1920x1080 visible upper",
                    kernel="halo_trace_kernel<0,3,true,kAccLogFinal,FISHEYE_EQUAL_AREA,UPPER,nogate> (regular-prism search, exit queue, hit log; last layer, lens, visible range and closed gate as template constants) + halo_split_kernel<1024,16,256> + halo_bin_accumulate_range_kernel",
                    metric="rays/sec (whole node) at 9 wavelengths, single-scatter hex column")
    if cfg == "2":
        return dict(scene=scenes.config3_scene(), render=scenes.config2_render(),
                    wls=[scenes.wl_discrete(w) for w in scenes.CONFIG_WAVELENGTHS_9], rays=50_000_000,
                    name="configs[2]: two-layer full multi-scattering (plate h=0.3 zenith gauss(0,0.8) prob 1.0 over random column h=1.3), 9 wavelengths x %d root rays per GPU per step, max_hits 7, fisheye_equal_area fov 180 1920x1080 visible upper",
                    kernel="halo_trace_kernel<0,3,true,kAccLogFinal,FISHEYE_EQUAL_AREA,UPPER,nogate> (transit source: layer 1 reads the continuation pool) + halo_split_kernel<1024,16,256> + halo_bin_accumulate_range_kernel",
                    metric="root rays/sec (whole node) at 9 wavelengths, two-layer full multi-scattering")
    if cfg in ("4", "4d", "4p"):
        full = {"type": "uniform", "mean": 0.0, "std": 360.0}
        g = {"type": "gauss", "mean": 1.0, "std": 0.15}
        if cfg == "4d":
            # the discrete reading: 31 spectrum entries 380..780 nm, weight = D65 SPD (GetIlluminantSpd, util/illuminant.cpp:113-134),
            # one session each (simulator.cpp:1099 loops over wl_params), rays per session = PerWavelengthRayNum = ceil(total / 31)
            import ctypes as C
            from ice_halo_sim_amd import backend
            L = backend.load_library()
            L.halo_host_illuminant_spd.restype, L.halo_host_illuminant_spd.argtypes = C.c_float, [C.c_int, C.c_float]
            lam = [380.0 + 400.0 * i / 30.0 for i in range(31)]
            wls = [scenes.wl_discrete(w, float(L.halo_host_illuminant_spd(abi.ILLUM["D65"], w))) for w in lam]
            return dict(scene=scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8),
                        render=scenes.render(abi.LENS_RECTANGULAR, 2048, 1024, el=0.0, visible=abi.VISIBLE_FULL),
                        wls=wls, rays=-(-25_000_000 // 31),
                        name="configs[4], discrete reading (primary, SURVEY 8(d) item 5): stochastic prism (examples/bench_config_stoch.json as shipped), full-sphere axis, "
                             "a 31-entry spectrum 380..780 nm weighted by the D65 SPD = 31 sessions x %d root rays per GPU per step (25 M per GPU, 200 M over 8 GPUs), max_hits 8, rectangular 2048x1024 visible full",
                        kernel="halo_trace_kernel<0,2,true,...> (sampled-prism pool, scalar plane of a discrete wavelength) + the route's accumulation passes (config.route)",
                        metric="rays/sec (whole node), stochastic-geometry crystal, 31 discrete wavelengths")
        if cfg == "4":
            e = scenes.stochastic_prism_entry()
            what = "stochastic prism (examples/bench_config_stoch.json as shipped: h=1, six gauss(1,0.15) face distances)"
            kern = "halo_trace_kernel<0,2,false,kAccLog> (X/Y/Z hit log) + halo_split_kernel<1024,16,512> + halo_log_accumulate_kernel<3>"
        else:
            e = scenes.entry(scenes.pyramid_crystal(0.1, 1.2, 0.5, upper_miller=(2, 3), face_distance=[g] * 6),
                             scenes.axis(zenith=full, azimuth=full, roll=full), 100.0, 5)
            what = "stochastic pyramid (config_example crystal 5, upper Miller (2,0,3), six gauss(1,0.15) face distances)"
            kern = "halo_trace_kernel<0,1,false,kAccLog> (X/Y/Z hit log) + halo_split_kernel<1024,16,512> + halo_log_accumulate_kernel<3>"
        return dict(scene=scenes.scene([(0.0, [e])], max_hits=8),
                    render=scenes.render(abi.LENS_RECTANGULAR, 2048, 1024, el=0.0, visible=abi.VISIBLE_FULL),
                    wls=[scenes.wl_illuminant("D65", 31)], rays=25_000_000,
                    name="configs[4]: " + what + ", full-sphere axis, D65 pool of 31 wavelengths, %d root rays per GPU per step (200 M over 8 GPUs), max_hits 8, rectangular 2048x1024 visible full",
                    kernel=kern, metric="rays/sec (whole node), stochastic-geometry crystal, 31 wavelengths")
    if cfg.startswith("ref:"):
        return ref_workload(cfg[4:])
    if cfg.startswith("filter:"):
        return filter_workload(cfg[7:])
    raise SystemExit("unknown --config %r" % cfg)


FILTER_CASES = ("none", "raypath_P", "entry_exit_PBD", "direction_out", "complex_PBD", "crystal_all_pass")


def filter_workload(name):
    """configs[1] with an emit-gate filter on its crystal entry (the filters of tools/perf_probe.py / tests' _filter_table): what the
    production filter kernels (kModeFilter) cost next to the plain kernel of the same scene.  `none` = configs[1] itself."""
    from ice_halo_sim_amd import scenes
    T = scenes.filter_term
    table = {"raypath_P": scenes.simple_filter(T("raypath", raypath=[3, 5]), "P"),
             "entry_exit_PBD": scenes.simple_filter(T("entry_exit", entry=3, exit=5, min_len=2, max_len=4), "PBD"),
             "direction_out": scenes.simple_filter(T("direction", az=180, el=20, radii=2.0), "", "filter_out"),
             "complex_PBD": scenes.complex_filter([[T("raypath", raypath=[1, 3, 2])], [T("entry_exit", entry=1), T("crystal", crystal_id=3)],
                                                   [T("raypath", raypath=[3, 1, 5, 7, 4])]], "PBD"),
             "crystal_all_pass": scenes.simple_filter(T("crystal", crystal_id=99), "", "filter_out")}
    if name != "none" and name not in table:
        raise SystemExit("unknown filter case %r (one of %s)" % (name, ", ".join(FILTER_CASES)))
    wk = workload("1")
    col = scenes.column_crystal_entry()
    col.filter_id = 0 if name == "none" else 1
    wk.update(scene=scenes.scene([(0.0, [col])], max_hits=7), filters=[] if name == "none" else [table[name]],
              name="filter:%s — configs[1]'s scene with %s on its crystal entry, " % (name, "no filter" if name == "none" else "the emit-gate filter `%s`" % name) +
                   "9 wavelengths x %d root rays per GPU per step, max_hits 7, fisheye_equal_area fov 180 1920x1080 visible upper",
              kernel="halo_trace_kernel<kModeFilter,3,true,kAccLogFinal,FISHEYE_EQUAL_AREA,UPPER,nogate> + split + per-tile sums" if name != "none" else wk["kernel"],
              metric="rays/sec, configs[1] under the emit-gate filter %s" % name)
    return wk


REF_SCENES = ("bench_light_single_ms", "ms_multi_crystal", "ms_multi_crystal_complex_filter", "ms_multi_crystal_filtered_bd")


def ref_workload(name):
    """A config document of the reference as a workload: the scene, the filters and the FIRST renderer of the document at its own
    resolution (examples/config_example.json: renderer 4, the one the survey's configs[0] names), its light source as written."""
    from ice_halo_sim_amd import config
    golden = os.path.join(ROOT, "tests", "golden")
    if name == "config_example":
        doc = json.load(open(os.path.join(golden, "ref_example_configs.json")))[name]
        where = "examples/config_example.json as shipped (README.md:46-48)"
    else:
        docs = json.load(open(os.path.join(golden, "ref_e2e_configs.json")))
        if name not in docs:
            raise SystemExit("unknown reference document %r" % name)
        doc = docs[name]
        where = "test/e2e/configs/%s.json (doc/performance-testing.md:465-468)" % name
    job = config.load_config(doc)
    rid = 4 if name == "config_example" else sorted(job.renders)[0]
    rd = job.renders[rid]
    if name == "config_example":
        wls, rays = list(job.wavelengths), job.per_wavelength_ray_num()      # 9 x 50 M
        what = "%d wavelengths x %%d root rays per GPU per step" % len(wls)
    else:
        wls, rays = list(job.wavelengths) * 4, 50_000_000                     # the document's light source, 4 sessions x 50 M
        what = "%d sessions x %%d root rays per GPU per step (light source as written: %s)" % (len(wls), "illuminant pool" if job.wavelengths[0].illuminant >= 0 else "discrete")
    filt = sum(1 for l in range(job.scene.layer_count) for e in range(job.scene.layers[l].entry_count) if job.scene.layers[l].entries[e].filter_id > 0)
    return dict(scene=job.scene, render=rd, wls=wls, rays=rays, filters=job.filters,
                colors=(job.color_sets, job.color_classes) if job.color_classes else None, geom_clock=job.geom_clock,
                name="ref:%s — %s, renderer %d (%dx%d), %d scattering layer(s), %d filtered crystal entr%s, max_hits %d, " % (
                    name, where, rid, rd.width, rd.height, job.scene.layer_count, filt, "y" if filt == 1 else "ies", job.scene.max_hits) + what,
                kernel="halo_trace_kernel (instantiations by route: see config.route) + its accumulation passes",
                metric="root rays/sec, reference document %s" % name)


def physical_cores():
    """physical cores of the host (SURVEY 8(d) / BASELINE.md 4 ask for the count to be stated): distinct (physical id, core id) pairs of
    /proc/cpuinfo; falls back to the logical count"""
    pairs, sockets, model = set(), set(), ""
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif line.startswith("model name") and not model:
                model = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                    sockets.add(phys)
                phys = core = None
    except OSError:
        pass
    return {"cores": len(pairs) or (os.cpu_count() or 1), "sockets": len(sockets) or 1, "model": model}


def cpu_baseline(wk, budget_s=15.0):
    """CPU oracle ("port") on all host cores, bounded to ~budget_s of work on the same workload shape."""
    from tests._oracle_backend import OracleBackend, run_session
    cores = os.cpu_count() or 1
    threads = min(cores, 128)
    physical = physical_cores()
    sc, rd, wls = wk["scene"], wk["render"], wk["wls"]
    ob = OracleBackend(seed=42, threads=threads, acc64=1)   # per-thread pixel caches: the shared float image's contended atomics kept the all-cores run from scaling
    if wk.get("geom_clock"):
        ob.set_option("geom_clock", wk["geom_clock"])
    if wk.get("filters"):
        ob.set_filters(wk["filters"])
    t0 = time.perf_counter()
    run_session(ob, sc, rd, wls[0], 20_000)   # warms the LUT / page cache / thread pool
    t0 = time.perf_counter()
    n_cal = 500_000 * max(1, threads // 8)    # calibration: ~0.6 s on the GPU box (the 0.1 s sample of round 5 put the timed sample anywhere between 7 and 35 s)
    run_session(ob, sc, rd, wls[0], n_cal)
    rate = n_cal / max(time.perf_counter() - t0, 1e-6)
    per_wl = int(max(20_000, min(wk["rays"], rate * budget_s / len(wls))))
    t0 = time.perf_counter()
    for wl in wls:
        run_session(ob, sc, rd, wl, per_wl)
    dt = time.perf_counter() - t0
    ob.close()
    return {"value": len(wls) * per_wl / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "host": {"threads_used": threads, "logical_cpus": cores, "physical_cores": physical["cores"], "sockets": physical["sockets"], "model": physical["model"],
                     "note": "`cores` above = OpenMP threads used (the contract's field); physical_cores = sockets x cores per socket from /proc/cpuinfo"},
            "sample": "same workload shape, %d session(s) x %d root rays (%.1f s of CPU work), OpenMP over rays in dynamic chunks of 4096 (the reference's protocol dispatches 128 rays per worker task, doc/performance-testing.md:109: the coarser chunk only flatters this number)" % (len(wls), per_wl, dt),
            "note": "kind=port: the reference's own CPU path (Simulator / CpuTraceBackend) needs spdlog + nlohmann-json >= 3.4, absent from this image, "
                    "and may not be built against stand-ins; this is the repo's C restatement of it (oracle/halo_oracle.c). Calibration of the real "
                    "reference: %.2f M rays/s on %d threads in the %s; the oracle runs %.1fx (1 thread) / %.1fx (6 threads) the compiled reference there (%s), "
                    "so this value overstates the reference's own CPU path by about that factor" % (
                        SURVEY_REFERENCE_CPU["rays_per_s"] / 1e6, SURVEY_REFERENCE_CPU["threads"], SURVEY_REFERENCE_CPU["where"],
                        ORACLE_OVER_REFERENCE["threads_1"], ORACLE_OVER_REFERENCE["threads_6"], ORACLE_OVER_REFERENCE["source"]),
            "oracle_over_reference": ORACLE_OVER_REFERENCE}


def _pmc_mean(name, key, kernel="halo_trace_kernel"):
    """mean-per-dispatch value of counter `key` for `kernel` in a committed rocprofv3 PMC summary (tools/rocpd_summary.py)"""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    val = None
    for line in open(path):
        if kernel in line and (" " + key + " ") in line:
            val = float(line.split(" " + key + " ")[1].split()[0])
    return val


def _pmc_kernel_us(name, key, kernel="halo_trace_kernel"):
    """average duration (us) of the kernel whose counter row _pmc_mean(name, key) reads, from the kernel table of the same summary"""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    full, rows = None, {}
    for line in open(path):
        if kernel in line and (" " + key + " ") in line:
            full = line.split(" " + key + " ")[0].strip()
        elif not line.startswith(" ") and len(line.split()) >= 6:
            f = line.rsplit(None, 6)
            try:
                rows[f[0].strip()] = float(f[3])
            except ValueError:
                pass
    return rows.get(full) if full else None


GROUP_KERNELS = ("halo_trace_kernel", "halo_split_kernel", "halo_bin_accumulate_range_kernel", "halo_bin_accumulate_kernel", "halo_log_accumulate_kernel")


def shape_record_bytes(crystal, prism_records, samples=256):
    """Bytes of a sampled-crystal record that the path NEEDS, from the face / fan-triangle counts of `samples` instances built on the host
    (the same builder the device generator runs): the generator writes the rows a record uses — header 16 B, 16 B per face, 32 B per
    opposite-face slab, 36 + 16 B per fan triangle (corners; normal + area), the face / triangle index bytes — and the trace reads, per
    32 rays that share the crystal, all of that EXCEPT the corner rows, of which it needs one 36-byte row per ray (the entry pick's one
    triangle; never more than all of them).  Round 3 charged the record's full sizeof (1360 / 4112 B) twice."""
    import ctypes as C
    import numpy as np
    from ice_halo_sim_amd import abi, backend
    L = backend.load_library()
    faces, tris = [], []
    for idx in range(samples):
        sc9 = np.zeros(9, np.float32)
        L.halo_host_shape_scalars(C.byref(crystal), 42, idx, 0, sc9.ctypes.data_as(C.POINTER(C.c_float)))
        d = sc9[3:9].copy()
        g = abi.HaloGeomTables()
        if crystal.kind == abi.CRYSTAL_PRISM:
            L.halo_host_prism_geometry(abs(float(sc9[0])), d.ctypes.data_as(C.POINTER(C.c_float)), C.byref(g))
        else:
            L.halo_host_pyramid_geometry(crystal.wedge_upper_deg, crystal.wedge_lower_deg, abs(float(sc9[0])), abs(float(sc9[1])), abs(float(sc9[2])),
                                         d.ctypes.data_as(C.POINTER(C.c_float)), C.byref(g))
        faces.append(g.face_cnt)
        tris.append(g.tri_cnt)
    f, t = float(np.mean(faces)), float(np.mean(tris))
    rows = 16.0 + 16.0 * f + 32.0 * (f / 2.0) + 16.0 * t + t + 2.0 * f      # header, faces, slabs (<= faces / 2), triangle normal + area rows, index bytes
    return {"mean_faces": f, "mean_fan_triangles": t, "sizeof_record": 1360 if prism_records else 4112,
            "written_by_generator": rows + 36.0 * t, "read_by_trace_per_32_rays": rows + 36.0 * min(32.0, t)}


def _profile_tag(cfg):
    """prefix of the committed rocprofv3 summaries of a configuration (tools/collect_all_profiles.sh names them)"""
    if cfg.startswith("ref:"):
        return "%s_ref_%s" % (PROFILE_ROUND, cfg[4:])
    if cfg.startswith("filter:"):
        return "%s_filter_%s" % (PROFILE_ROUND, cfg[7:])
    return "%s_bench%s" % (PROFILE_ROUND, cfg)


def pmc_traffic_per_launch(cfg):
    """HBM-side bytes per launch GROUP — the trace kernel plus the accumulation passes that follow it on the stream (split,
    per-tile sums), which is also what the HIP events around a launch time — from the committed rocprofv3 PMC passes of this
    same command (profiles/<round>_bench<cfg>_pmc_{fetch,write}_size.txt; FETCH_SIZE and WRITE_SIZE are collected in their own
    passes — they do not fit one — and never inside the timed run).  Units and corrections per MI355X_MICROARCH.md §HBM: both
    counters are in KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming reads (64 B tallied per 128-B
    request), so it is doubled; WRITE_SIZE is uncalibrated and taken as reported.  Sum of the per-dispatch means of the
    group's kernels."""
    tag = _profile_tag(cfg)
    tot, seen = 0.0, False
    for k in GROUP_KERNELS:
        f, w = _pmc_mean(tag + "_pmc_fetch_size.txt", "FETCH_SIZE", k), _pmc_mean(tag + "_pmc_write_size.txt", "WRITE_SIZE", k)
        if f is not None and w is not None:
            tot += (2.0 * f + w) * 1024.0
            seen = seen or k == "halo_trace_kernel"
    return tot if seen else None


def _micro_costs_file():
    """the committed output of tools/valu_rate_bench.hip: this round's, else the newest round present (instruction costs are a property of
    the part, not of the tree; the file used is named in the line)"""
    rounds = [PROFILE_ROUND] + ["r%02d" % k for k in range(int(PROFILE_ROUND[1:]) - 1, 0, -1)]
    for r in rounds:
        path = os.path.join(ROOT, "profiles", "%s_micro_valu_rate_bench.txt" % r)
        if os.path.exists(path):
            return path
    return None


def _micro_costs():
    """SIMD cycles per wave64 instruction from the committed micro-benchmark output (profiles/<round>_micro_valu_rate_bench.txt,
    tools/valu_rate_bench.hip run on the GPU box): {instruction label: cycles}."""
    path = _micro_costs_file()
    out = {}
    if path:
        for line in open(path):
            if "cycles per wave64 instruction" in line:
                name = line[:38].strip()
                out[name] = float(line.split("ms")[1].split("cycles")[0])
    return out


PMC_PASSES = ("kernel_stats", "pmc_fetch_size", "pmc_write_size", "pmc_insts", "pmc_cycles", "pmc_classes", "pmc_wait")


def _full_counters(cfg):
    """BASELINE configurations carry the compute-side counter passes; the reference's documents only the kernel table and the HBM traffic pair"""
    return not cfg.startswith("ref:")


def profile_files_read(cfgs=None):
    """every file under profiles/ that a default `python bench.py` run reads (repo-relative): the seven summaries per configuration of
    tools/collect_profiles.sh and the micro-benchmark's output.  tests/test_bench_profiles.py fails when one is absent."""
    cfgs = cfgs if cfgs is not None else ("1",) + OTHER_CONFIGS
    out = ["profiles/%s_%s.txt" % (_profile_tag(c), p) for c in cfgs for p in (PMC_PASSES if _full_counters(c) else PMC_PASSES[:3])]
    out.append("profiles/%s_micro_valu_rate_bench.txt" % PROFILE_ROUND)
    return out


def rocprof_kernel_avg(cfg, kernel="halo_trace_kernel"):
    """(average, minimum, calls, average without the first call's share) in ms of the dominant trace kernel in the committed rocprofv3
    --kernel-trace --stats summary of this configuration (profiles/<round>_bench<cfg>_kernel_stats.txt): the LAST listed instantiation with
    the most total time, like _pmc_mean."""
    path = os.path.join(ROOT, "profiles", _profile_tag(cfg) + "_kernel_stats.txt")
    if not os.path.exists(path):
        return None
    best = None
    for line in open(path):
        if kernel in line and not line.startswith(" "):
            f = line.rsplit(None, 6)
            try:
                calls, total, avg, mn, mx = int(f[1]), float(f[2]), float(f[3]), float(f[4]), float(f[5])
            except (ValueError, IndexError):
                continue
            if best is None or total > best["total_us"]:
                best = {"kernel": f[0].strip(), "calls": calls, "total_us": total, "avg_ms": avg / 1e3, "min_ms": mn / 1e3, "max_ms": mx / 1e3,
                        "avg_ms_without_slowest_call": (total - mx) / max(calls - 1, 1) / 1e3, "file": "profiles/" + os.path.basename(path)}
    return best


def pmc_valu(cfg, rays_per_launch):
    """Compute-side reading of the trace kernel from the committed PMC passes (per-dispatch means are per counter instance = one shader
    engine = 8 CUs = 32 SIMDs).
      issue_frac = sum over instruction classes of (dynamic count x measured cost of the class) / (32 SIMDs x SQ_BUSY_CYCLES): the share of
        the SIMDs' cycles that the kernel's own VALU instructions need at the rates tools/valu_rate_bench.hip measures on this part (an
        fp32 FMA is 2.8 cycles per wave64 instruction, a v_cndmask or v_cmp 4.5-4.8, an integer multiply 4.8, a transcendental 8.4) —
        counts by class from SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 / INT32 / INT64 / CVT, the unclassified rest (moves, selects, compares,
        logic, lane moves) priced at the static mix of the kernel's hot loop.
      The older figure, SQ_ACTIVE_INST_VALU x 4 / 32 / SQ_BUSY_CYCLES, is kept as valu_active_frac: it charges every instruction four
        cycles and reads slightly above 1 when the pipe is saturated."""
    tag = _profile_tag(cfg)
    insts = _pmc_mean(tag + "_pmc_insts.txt", "SQ_INSTS_VALU")
    cyc = tag + "_pmc_cycles.txt"
    active, busy, wave = _pmc_mean(cyc, "SQ_ACTIVE_INST_VALU"), _pmc_mean(cyc, "SQ_BUSY_CYCLES"), _pmc_mean(cyc, "SQ_WAVE_CYCLES")
    if not insts or not active or not busy:
        return None
    out = {"valu_active_frac": active * 4.0 / 32.0 / busy, "valu_insts_per_wave_ray": insts * 32.0 / (rays_per_launch / 64.0),
           "waves_per_simd": (wave * 4.0 / busy / 32.0) if wave else None,
           "source": "profiles/%s_pmc_{insts,cycles,classes,wait}.txt (counter means over every launch of the kernel in that pass)" % tag}
    cls = tag + "_pmc_classes.txt"
    keys = ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT")
    counts = {k: _pmc_mean(cls, "SQ_INSTS_VALU_" + k) for k in keys}
    total = _pmc_mean(cls, "SQ_INSTS_VALU")
    costs = _micro_costs()
    if costs:
        out["instruction_costs_file"] = "profiles/" + os.path.basename(_micro_costs_file())
    if total and all(v is not None for v in counts.values()) and costs:
        c = lambda name, default: costs.get(name, default)
        fma = c("v_fma_f32", 2.81)
        price = {"ADD_F32": fma, "MUL_F32": fma, "FMA_F32": fma,
                 "TRANS_F32": (c("v_rcp_f32", 8.6) + c("v_sqrt_f32", 8.3) + c("v_rsq_f32", 8.2)) / 3.0,
                 # integer adds / shifts / logic next to integer multiplies, 3 : 1 in the PCG hash that dominates this class
                 "INT32": 0.75 * c("v_add_u32", 3.2) + 0.25 * c("v_mul_lo_u32", 4.8), "INT64": c("v_mad_u64_u32", 5.9), "CVT": c("v_cvt_i32_f32", 4.2)}
        # the unclassified instructions, by the static mix of the interaction loop (v_cndmask 27 %, v_cmp 22 %, v_mov 22 %, logic / shifts 19 %, lane moves 10 %)
        other = 0.27 * c("v_cndmask_b32_e64 (sgpr mask)", 4.45) + 0.22 * c("v_cmp_gt_f32", 4.86) + 0.22 * c("v_mov_b32", 2.65) + \
            0.19 * c("v_xor_b32", 3.1) + 0.10 * c("v_readlane_b32", 4.56)
        classified = sum(counts.values())
        cycles = sum(counts[k] * price[k] for k in keys) + max(total - classified, 0.0) * other
        # In TIME, not in counted cycles: the micro-benchmark's "cycles" are its elapsed time x 2.4 GHz, and the part clocks lower under this
        # kernel (SQ_BUSY_CYCLES / duration = 2.1 GHz in the round-4 passes) — so the priced instruction stream is compared with the kernel's
        # own duration in the pass that counted it.  issue_frac_cycles keeps round 3's cycle-domain figure for comparison (it reads ~12 % high).
        dur_us = _pmc_kernel_us(cls, "SQ_INSTS_VALU")
        out["issue_frac_cycles"] = cycles / (32.0 * busy)
        out["issue_frac"] = (cycles / 2.4e9 / 32.0) / (dur_us * 1e-6) if dur_us else out["issue_frac_cycles"]
        out["kernel_us_in_counter_pass"] = dur_us
        out["busy_cycles_per_us"] = busy / _pmc_kernel_us(cyc, "SQ_BUSY_CYCLES") if _pmc_kernel_us(cyc, "SQ_BUSY_CYCLES") else None
        out["census"] = {k: counts[k] / total for k in keys}
        out["census"]["other (moves, selects, compares, logic, lane moves)"] = max(total - classified, 0.0) / total
        out["cost_cycles_per_wave64_instruction"] = dict(price, other=other)
    w = tag + "_pmc_wait.txt"
    wa, wi, aa = _pmc_mean(w, "SQ_WAIT_ANY"), _pmc_mean(w, "SQ_WAIT_INST_ANY"), _pmc_mean(w, "SQ_ACTIVE_INST_ANY")
    wv = _pmc_mean(w, "SQ_WAVE_CYCLES")
    if wa and wi and aa and wv:
        out["wave_cycles"] = {"issuing": aa / wv, "ready_but_pipe_busy": wi / wv, "parked_at_waitcnt": wa / wv}
    out["note"] = "issue_frac ~ 0.9 with ready waves waiting for the pipe a quarter of the time = VALU-issue-bound; valu_active_frac (4 cycles per instruction) reads above 1 then"
    return out


def measure(cfg, args, ctx, steps, warmup, repeats, with_cpu):
    """One configuration on this rank's GPU: `warmup` untimed steps, then `repeats` timed regions of exactly `steps` steps each (barrier +
    synchronize on both sides, MAX over ranks).  Returns the JSON object of the configuration on rank 0, None elsewhere."""
    torch, dist = ctx["torch"], ctx["dist"]
    world, rank, local_rank = ctx["world"], ctx["rank"], ctx["local_rank"]
    from ice_halo_sim_amd.dist import ShardedTracer

    wk = workload(cfg)
    sc, rd, wls = wk["scene"], wk["render"], wk["wls"]
    n_cfg = args.rays_per_wl or wk["rays"]
    strong = args.scaling == "strong"
    n = -(-n_cfg // world) if strong else n_cfg        # strong: the configuration's rays are the whole job, cut into `world` shards
    tracer = ShardedTracer(sc, rd, seed=42, device=local_rank, rank=rank, world=world, **{"async": 1})
    if wk.get("geom_clock"):
        tracer.backend.set_option("geom_clock", wk["geom_clock"])
    if wk.get("filters"):
        tracer.backend.set_filters(wk["filters"])
    if wk.get("colors"):
        tracer.backend.set_color(*wk["colors"])
    if args.blocks_per_cu > 0:
        tracer.backend.set_option("blocks_per_cu", args.blocks_per_cu)
    if args.aggregate >= 0:
        tracer.backend.set_option("aggregate", args.aggregate)
    for kv in args.opt:
        k, v = kv.split("=")
        tracer.backend.set_option(k, int(v))
    layers = sc.layer_count
    first_layer = {"ms": 0.0, "launches": 0, "hits": 0, "cont": 0}     # non-final layers return their tallies synchronously

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    reduce_events = []                                       # in-line mode only: (start, end) torch events around each step's collective on the trace stream

    def step():
        for wl in wls:
            sts = tracer.trace_session_layers(wl, n)        # this rank's shard of the batch
            for st in sts[:-1]:
                first_layer["ms"] += st.kernel_ms
                first_layer["launches"] += st.launches
                first_layer["hits"] += st.pixel_hits
                first_layer["cont"] += st.continuation_count
        if world > 1 and not tracer.two:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            tracer.reduce_to_root()                          # one RCCL sum-reduce at the drain point, in line with the trace kernels
            e1.record()
            reduce_events.append((e0, e1))
        else:
            tracer.reduce_to_root()                          # N > 1: queued on the side stream, the next step traces into the other tensor (dist.ShardedTracer)

    for _ in range(warmup):
        step()
    if args.warm_seconds > 0.0:
        # Short steps (config 4d: 31 sessions of 0.1 ms kernels) are over before the device has settled — its first timed steps read 13.8 / 9.7 /
        # 7.8 ms alone, and 12-15 ms for half a second right behind a configuration that had gigabytes of pools to give back (configs 2, 4).
        # Untimed, like the steps above: keep stepping in chunks of `steps` until two consecutive chunks agree within 5 % (at least
        # warm_seconds, at most 3 s).
        t_w, last = time.perf_counter(), None
        while True:
            t_c = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            now = time.perf_counter()
            chunk = now - t_c
            settled = last is not None and abs(chunk - last) <= 0.05 * chunk
            last = chunk
            if (settled and now - t_w >= args.warm_seconds) or now - t_w > 3.0:
                break
    tracer.zero()
    tracer.backend.collect_stats()                            # drop the warm-up tallies
    tracer.backend.collect_timing()
    for k in first_layer:
        first_layer[k] = 0
    reduce_events.clear()
    del tracer.reduce_spans[:]

    def timed_region():
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()                                           # dispatches are queued; nothing waits on the host per launch
        if world > 1:
            tracer._join()                                   # the last drains' collectives are part of the steps that queued them
        barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # N > 1: which drain?  dist.ShardedTracer can queue the collective on a side stream (the next step traces into a second tensor meanwhile)
    # or keep it in line.  The side stream wins when every rank has its own GPU and the collective is a device-side RCCL kernel; it LOSES in
    # the one-GPU gloo rehearsal, where N processes with deep queues time-slice one device and every all_reduce needs all of them to get a
    # turn (profiles/r06_gloo_rehearsal.txt).  So the mode is chosen by measurement, outside the timed region: two regions of each after the
    # warm-up, the faster one ships (all ranks see the same MAX-reduced times, so they agree), and the line reports both.
    drain_probe = None
    if world > 1 and len(tracer.accs) == 2:
        drain_probe = {}
        for mode in (True, False):
            tracer.set_overlap(mode)
            drain_probe[mode] = min(timed_region() for _ in range(2))
        tracer.set_overlap(drain_probe[True] <= 1.02 * drain_probe[False])
        tracer.backend.collect_stats()
        tracer.backend.collect_timing()
        for k in first_layer:
            first_layer[k] = 0
        reduce_events.clear()
        del tracer.reduce_spans[:]
    times = [timed_region() for _ in range(max(repeats, 1))]
    drain_switched = False
    if drain_probe is not None and tracer.two and statistics.median(times) > 1.25 * drain_probe[False]:
        # Safety net.  The probe's two short regions per mode do not always show what the side-stream drain does once the queues are deep: in the
        # one-GPU gloo rehearsal it probed 43.9 vs 43.0 ms per step, was chosen (within 2 %), and then ran 451 ms per step.  When the chosen
        # side-stream drain runs a quarter slower than the in-line drain PROBED, the line is measured again with the in-line drain and says so
        # (every rank sees the same MAX-reduced times, so they all switch).
        tracer._join()
        tracer.set_overlap(False)
        tracer.backend.collect_stats()
        tracer.backend.collect_timing()
        for k in first_layer:
            first_layer[k] = 0
        reduce_events.clear()
        del tracer.reduce_spans[:]
        times = [timed_region() for _ in range(max(repeats, 1))]
        drain_switched = True
    trace_ms, post_ms, timed_launches = tracer.backend.collect_timing()   # HIP events of the LAST layer's launches: the trace kernels' own spans / their accumulation passes'
    st = tracer.backend.collect_stats()                      # HIP-event kernel times + device tallies of every timed repeat
    route = tracer.backend.last_route()
    overlap_ab = None
    if drain_probe is not None:
        spans = [a.elapsed_time(b) for a, b, _, _ in tracer.reduce_spans]
        overlap_ab = {"drain_chosen": "side_stream" if tracer.two else "in_line", "switched_after_the_timed_run": drain_switched,
                      "probe_ms_per_step": {"side_stream": drain_probe[True] * 1e3 / steps, "in_line": drain_probe[False] * 1e3 / steps},
                      "reduce_ms_events_mean": (sum(spans) / len(spans)) if spans else None, "reduces_timed": len(spans),
                      "note": "probe = the faster of two untimed regions per mode, after the warm-up; `value` is measured in the chosen mode. reduce_ms: events on the "
                              "side stream around the queued collective and the drain of the other ranks (under gloo with the side-stream drain the end event is "
                              "recorded when this rank next needs the tensor)"}
    dt = statistics.median(times)
    cov = (statistics.pstdev(times) / statistics.mean(times)) if len(times) > 1 else 0.0

    reps = len(times)
    rays_per_rank_step = len(wls) * n
    rays_per_rank = reps * steps * rays_per_rank_step
    launches, kernel_ms, pixel_hits, exits = int(st.launches), float(st.kernel_ms), int(st.pixel_hits), int(st.exit_count)
    # the DEVICE's own root tally must equal the rays claimed (nothing skipped inside the timed region)
    assert int(st.root_count) == rays_per_rank + first_layer["cont"], (int(st.root_count), rays_per_rank, first_layer["cont"])

    img, landed = tracer.readback()                          # collective: landed-weight scalars are summed here, once
    multi = certify_multi_gpu(ctx, tracer, wls[0], min(n, 4_000_000), reduce_events, dist_backend=ctx.get("dist_backend", "nccl")) if world > 1 else None
    if multi is not None and overlap_ab is not None:
        multi["reduce_overlap"] = overlap_ab
    tracer.backend.close()
    del tracer
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    # the dominant kernel: the last layer's launches (single-scatter: the only layer; multi-scatter: the transit-source layer)
    dom_launches = launches - first_layer["launches"]
    # the dominant KERNEL: the last layer's trace kernel, timed by the HIP events around it on its own stream (halo_collect_timing counts the
    # last layers' launches only); group_ms adds its accumulation passes (split, per-tile sums), which follow it on that stream
    # (a launch whose HIP event pair read backwards is left out of the timing sums by the library — halo_backend.cpp harvest_slot — so the averages
    # below run over the launches that WERE timed; the line says how many)
    assert 0 < timed_launches <= dom_launches, (timed_launches, dom_launches)
    timed_frac = timed_launches / dom_launches
    dom_ms = trace_ms / timed_frac
    group_ms = kernel_ms - first_layer["ms"]
    dom_hits = pixel_hits - first_layer["hits"]
    dom_rays = (first_layer["cont"] if layers > 1 else rays_per_rank)
    avg_launch_s = dom_ms * 1e-3 / max(dom_launches, 1)
    # ALGORITHMIC bytes (the fused definition): 3 ch x 4 B x (read + write) per in-frame pixel hit; a
    # transit-source launch also reads its 20-byte continuation records (and the layer before wrote them: 20 B out + 20 B in per continuation).
    # What a shape-pool launch moves for its sampled-crystal records (the generator writes one per 32 rays, the trace reads it) exists only
    # because THIS design stages shapes through HBM: it is reported beside the roofline as design traffic, not priced as algorithmic.
    alg = dom_hits * 24.0
    if layers > 1:
        alg += dom_rays * 40.0
    shape_bytes, design = None, 0.0
    if cfg in ("4", "4d", "4p"):
        shape_bytes = shape_record_bytes(sc.layers[0].entries[0].crystal, prism_records=(cfg != "4p"))
        design = dom_rays / 32.0 * (shape_bytes["written_by_generator"] + shape_bytes["read_by_trace_per_32_rays"])
    alg_per_launch = alg / max(dom_launches, 1)
    design_per_launch = design / max(dom_launches, 1)
    achieved = alg_per_launch / max(avg_launch_s, 1e-12) / 1e9
    valu = pmc_valu(cfg, dom_rays / max(dom_launches, 1)) if _full_counters(cfg) else None   # counters of the dominant layer's kernel (the last listed instantiation)
    rp = rocprof_kernel_avg(cfg)
    total_rays_step = rays_per_rank_step * world
    out = {
        "metric": wk["metric"],
        "value": total_rays_step * steps / dt,
        "unit": "rays/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": dt * 1e3 / steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wk["name"] % n,
                   "rays_per_step_per_gpu": rays_per_rank_step, "resolution": [rd.width, rd.height],
                   "sharding": "root-ray index ranges, 1 RCCL reduce of %d floats per step" % (rd.width * rd.height * 3 + 4),
                   "exits_per_root": exits / max(rays_per_rank, 1), "landed_weight_rank0_image": landed,
                   "route": {"mode_mask": route.mode_mask, "geom_mask": route.geom_mask, "accum_mask": route.accum_mask,
                             "source_mask": route.source_mask, "planes": route.plane_cnt, "plane_copies": route.plane_copies,
                             "spec_mask": route.spec_mask, "generic_launches": route.generic_launches}},
        "repeats": {"n": reps, "protocol": "reference doc/performance-testing.md:109-131 (>= 5 repeats, median + CoV); each repeat times exactly --steps steps",
                    "median_ms_per_step": dt * 1e3 / steps, "cov": cov,
                    "ms_per_step_all": [x * 1e3 / steps for x in times]},
        "roofline": {"bound": "valu_issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_per_launch(cfg),
                     "priced_against": "hbm: achieved / peak / frac are the contract's HBM reading (algorithmic bytes per launch / the trace kernel's mean duration / 8 TB/s); `bound` names the roof that binds — VALU issue, see `binding`",
                     "hbm_frac": achieved / HBM_PEAK_GBS,
                     "binding": {"roof": "valu_issue", "frac": (valu or {}).get("issue_frac"),
                                 "meaning": "share of the kernel's own duration that its VALU instruction stream needs at the per-class issue costs measured on this part (`valu`); ~0.9-1.0 = the SIMDs' issue slots are the limit, not memory"},
                     "traffic_source": "profiles/%s_pmc_{fetch,write}_size.txt (separate rocprofv3 --pmc passes of this command; bytes per launch group = sum over the trace kernel and its split / per-tile-sum passes of (2 x FETCH_SIZE + WRITE_SIZE) KB, per-dispatch means: gfx950 FETCH_SIZE counts half of coalesced reads, WRITE_SIZE uncalibrated)" % _profile_tag(cfg),
                     "kernel": wk["kernel"], "launches": dom_launches,
                     "avg_launch_ms": avg_launch_s * 1e3, "avg_launch_group_ms": group_ms / max(dom_launches, 1), "passes_ms_per_launch": post_ms / max(timed_launches, 1), "timed_launches": timed_launches,
                     "wall_ms_per_launch": dt * 1e3 / steps / max(dom_launches / (reps * steps), 1),
                     "algorithmic_bytes_per_launch": alg_per_launch,
                     "design_bytes_per_launch": design_per_launch if shape_bytes else None,
                     "frac_with_design_bytes": ((alg_per_launch + design_per_launch) / max(avg_launch_s, 1e-12) / 1e9 / HBM_PEAK_GBS) if shape_bytes else None,
                     "shape_record_bytes": shape_bytes,
                     "kernel_rays_per_s": dom_rays / max(dom_ms * 1e-3, 1e-12),
                     "rocprof": None if rp is None else dict(rp, frac_at_rocprof_avg=alg_per_launch / (rp["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                             frac_at_rocprof_avg_without_slowest_call=alg_per_launch / (rp["avg_ms_without_slowest_call"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                             hip_events_over_rocprof_avg=avg_launch_s * 1e3 / rp["avg_ms"]),
                     "valu": valu,
                     "note": "avg_launch_ms = the trace kernel alone, HIP events on its stream inside THIS run's timed region (warm launches only). `rocprof` = the same kernel in the committed rocprofv3 --kernel-trace --stats summary of this command (every call of a short profiled run, its first cold call included; kernels run a few per cent slower under the profiler): the two durations and both fractions are given, hip_events_over_rocprof_avg says by how much they differ. The kernel's accumulation passes (split + per-tile sums) follow it on the same stream (avg_launch_group_ms = both spans added, wall_ms_per_launch = the timed region / launches: what is left over is the sampled-crystal generator, the closing fold and launch gaps). The fused kernel keeps rays in registers: HBM sees hit records / accumulator RMWs (+ continuation records; + the sampled-crystal records of a shape-pool launch, reported as design_bytes_per_launch and NOT priced as algorithmic), so the path is VALU-issue-bound, not HBM-bound (DESIGN.md §4)"},
    }
    if out["roofline"]["traffic"]:
        out["roofline"]["traffic_over_algorithmic"] = out["roofline"]["traffic"] / alg_per_launch
        if shape_bytes:
            out["roofline"]["traffic_over_algorithmic_plus_design"] = out["roofline"]["traffic"] / (alg_per_launch + design_per_launch)
    if layers > 1:
        traced = rays_per_rank + first_layer["cont"]
        out["multi_scatter"] = {"root_rays_per_s": out["value"], "traced_rays_per_s": traced / (reps * steps) * world / (dt / steps),
                                "continuations_per_root": first_layer["cont"] / max(rays_per_rank, 1),
                                "first_layer_kernel_ms_per_launch": first_layer["ms"] / max(first_layer["launches"], 1),
                                "last_layer_kernel_ms_per_launch": avg_launch_s * 1e3}
    if multi is not None:
        out["multi_gpu"] = multi
    if with_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(wk)
    return out


def certify_multi_gpu(ctx, tracer, wl, n_check, reduce_events, dist_backend):
    """What makes an N > 1 line prove itself (every rank calls this; rank 0 gets the object): how many ranks the process group saw and on
    which devices, what the drain-point collective cost (torch events around every timed step's reduce, per rank), and a check pass —
    each rank traces `n_check` more roots, the ranks' OWN images (before the reduce) must differ from one another (disjoint ray-counter
    ranges: a replica would be bit-identical) while their energies agree to 1 %, and the image the collective leaves on rank 0 must
    be their sum."""
    import numpy as np
    torch, dist = ctx["torch"], ctx["dist"]
    world, rank, local_rank = ctx["world"], ctx["rank"], ctx["local_rank"]
    torch.cuda.synchronize()
    reduce_ms = [a.elapsed_time(b) for a, b in reduce_events]   # the in-line pass's events (on the trace stream: they include the wait for the slowest rank)
    prop = torch.cuda.get_device_properties(local_rank)
    me = {"rank": rank, "local_rank": local_rank, "device_index": torch.cuda.current_device(), "name": prop.name,
          "uuid": str(getattr(prop, "uuid", "")), "pci": "%04x:%02x:%02x" % (getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_bus_id", 0), getattr(prop, "pci_device_id", 0)),
          "hostname": os.uname().nodename, "pid": os.getpid(),
          "reduce_ms_mean": (sum(reduce_ms) / len(reduce_ms)) if reduce_ms else None, "reduce_ms_max": max(reduce_ms) if reduce_ms else None, "reduces_timed": len(reduce_ms)}
    # the check pass: this rank's own image first ...
    tracer.zero()
    tracer.trace_session(wl, n_check)
    tracer.backend.flush()                                   # (the closing fold is deferred: bring the bound tensor up to date before reading it)
    torch.cuda.synchronize()
    w, h = tracer.render.width, tracer.render.height
    own = tracer.acc[: w * h * 3].double()
    gen = torch.Generator(device="cpu").manual_seed(1234)
    probe = torch.rand(4096, generator=gen, dtype=torch.float64).to(own.device)      # a fixed random functional: two different images do not share it
    idx = torch.randint(0, w * h * 3, (4096,), generator=gen).to(own.device)
    me["check_sum_y"] = float(own[1::3].sum().item())
    me["check_signature"] = float((own[idx] * probe).sum().item())
    me["check_landed"] = float(tracer.backend.take_landed())
    # ... then the collective, and what it leaves on the root
    tracer.reduce_to_root()
    held = tracer.total()                                    # behind the queued reduce: the root's running total, nothing on the others
    torch.cuda.synchronize()
    reduced_y = float(held[: w * h * 3].double()[1::3].sum().item())
    nonroot_drained = bool(rank == 0 or not held.any().item())
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(me, nonroot_drained=nonroot_drained))
    tracer.zero()
    if rank != 0:
        return None
    ys = np.array([g["check_sum_y"] for g in gathered])
    sigs = [g["check_signature"] for g in gathered]
    distinct = len(set(sigs)) == world
    agree = bool(np.all(np.abs(ys / ys.mean() - 1.0) <= 0.01))
    summed = bool(abs(reduced_y - ys.sum()) <= 1e-5 * ys.sum())
    devices = sorted({(g["hostname"], g["uuid"] or g["pci"], g["device_index"]) for g in gathered})
    out = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "dist_backend_requested": dist_backend,
           "distinct_devices": len(devices), "ranks": gathered,
           "reduce_ms_mean_over_ranks": float(np.mean([g["reduce_ms_mean"] for g in gathered if g["reduce_ms_mean"] is not None])) if any(g["reduce_ms_mean"] is not None for g in gathered) else None,
           "reduce_floats": w * h * 3 + 4,
           "check": {"rays_per_rank": n_check, "rank_images_differ": distinct, "rank_energies_within_1pct": agree, "reduced_image_is_the_sum": summed,
                     "nonroot_ranks_drained": all(g["nonroot_drained"] for g in gathered),
                     "sum_y_per_rank": ys.tolist(), "sum_y_reduced_on_root": reduced_y},
           "note": "ranks_seen / devices come from the process group and the runtime, not from --gpus; reduce_ms brackets the stream-ordered collective with events "
                   "(it includes waiting for the slowest rank's trace: ranks arrive at the collective when their own kernels finish)"}
    assert distinct and agree and summed and out["check"]["nonroot_ranks_drained"], out["check"]
    if dist_backend == "nccl":
        assert len(devices) == world, ("ranks share a device under RCCL", devices)
    return out


OTHER_CONFIGS = ("2", "4", "4d", "4p") + tuple("ref:" + n for n in REF_SCENES) + ("ref:config_example",)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <a free port> bench.py <same arguments>`) and return its exit code.  Under
    RCCL every rank needs its own device, so fewer visible devices than N is refused here, before any rank starts (HALO_BENCH_BACKEND=gloo
    lets ranks share devices: the rehearsal of the N > 1 path on a one-GPU box)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("HALO_BENCH_BACKEND", "nccl") == "nccl" and have < n:
        sys.stderr.write("bench.py --gpus %d: only %d HIP device(s) visible (one rank per GPU under RCCL)\n" % (n, have))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="1", help="1 | 2 | 4 | 4d | 4p | ref:<document> (see the module docstring)")
    ap.add_argument("--repeats", type=int, default=5, help="how many times the timed region of --steps steps is run (median reported)")
    ap.add_argument("--rays-per-wl", type=int, default=0, help="root rays per session per GPU per step (0 = the configuration's own)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: per-GPU work fixed; strong: the configuration's rays are the whole job")
    ap.add_argument("--warm-seconds", type=float, default=0.0, help="extra untimed warm-up after the --warmup steps: chunks of --steps steps until two consecutive chunks agree within 5 %% (at least this long, at most 3 s; the short runs of the other configurations use 0.3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="default configuration on one GPU: skip the short runs of the other configurations")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--aggregate", type=int, default=-1)
    ap.add_argument("--opt", action="append", default=[], help="backend option key=value (repeatable)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as plain `python bench.py --gpus N`: become the launcher — one rank per GPU under torch.distributed.run, this
        # process only relays rank 0's JSON line and the exit code
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the product has no CPU path)")
    # HALO_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N>1 path on a 1-GPU box); default RCCL
    dist_backend = os.environ.get("HALO_BENCH_BACKEND", "nccl")
    if dist_backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=dist_backend)
    ctx = {"torch": torch, "dist": dist, "world": world, "rank": rank, "local_rank": local_rank, "dist_backend": dist_backend}

    out = measure(args.config, args, ctx, args.steps, args.warmup, args.repeats, with_cpu=not args.no_cpu_baseline)
    if args.config == "1" and world == 1 and not args.no_others and not args.rays_per_wl:
        others = {}
        args.warm_seconds = max(args.warm_seconds, 0.3)
        for cfg in OTHER_CONFIGS:
            r = measure(cfg, args, ctx, 3, 3, 3, with_cpu=False)   # (3 warm-up steps: a step of 31 short sessions is over before the clocks have ramped — config 4d read 13.8 / 9.7 / 7.8 ms on its first three steps)
            others[cfg] = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": 3, "warmup": 3,
                           "repeats": r["repeats"]["n"], "cov": r["repeats"]["cov"],
                           "workload": r["config"]["workload"], "resolution": r["config"]["resolution"], "exits_per_root": r["config"]["exits_per_root"],
                           "route": r["config"]["route"],
                           "roofline": {k: r["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "avg_launch_group_ms", "launches", "kernel_rays_per_s",
                                                                                  "traffic", "traffic_over_algorithmic", "algorithmic_bytes_per_launch", "design_bytes_per_launch",
                                                                                  "frac_with_design_bytes")}}
            others[cfg]["roofline"]["valu_issue_frac"] = (r["roofline"].get("valu") or {}).get("issue_frac")
            if "multi_scatter" in r:
                others[cfg]["multi_scatter"] = r["multi_scatter"]
        out["other_configs"] = others
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
