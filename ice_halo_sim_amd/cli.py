"""Command-line front end for the HIP trace backend: run a Lumice JSON config through the path and, with --benchmark,
print the reference's one-line `[BENCHMARK] {json}` record (reference src/main.cpp:342-573; rate definition
doc/performance-testing.md:86-131: Σ root rays over wavelengths / steady seconds, setup excluded).

  python -m ice_halo_sim_amd.cli -f examples/config_example.json --render 4 --benchmark
"""
import argparse
import json
import os
import sys
import time

import numpy as np

from . import config
from .backend import BackendUnavailableError, HipTraceBackend

DISPATCH_RAYS = 1 << 26  # rays per TraceLayer call (the reference's GPU dispatch is 2^18, server.cpp:151; we batch far larger)
DISPATCH_RAYS_MULTI = 1 << 24  # ... for scenes with more than one scattering layer: the continuation pools are sized for roots x max_hits,
                               # and a one-shot CLI run pays for their allocation (64 Mi roots x 8 hits: two pools of 10 GB, 0.5 s of hipMalloc)


def run_job(job, render_id=None, seed=42, device=0, max_rays=None, progress=None):
    """Trace `job` (config.TraceJob) on one GPU. Returns dict(rays, setup_sec, active_sec, backend, render)."""
    if not job.renders:
        raise config.ConfigError("config has no render entry")
    rid = render_id if render_id is not None else sorted(job.renders)[0]  # the seam supports ONE renderer (simulator.cpp:937-944)
    render = job.renders[rid]
    t0 = time.perf_counter()
    be = HipTraceBackend(device=device, seed=seed)
    if job.geom_clock:
        be.set_option("geom_clock", job.geom_clock)
    be.set_filters(job.filters)
    if job.color_classes:
        be.set_color(job.color_sets, job.color_classes)
    total = job.ray_num if job.ray_num is not None else (max_rays or 0)
    if max_rays is not None:
        total = min(total, max_rays)
    n_wl = max(1, len(job.wavelengths))
    per_wl = -(-total // n_wl)
    # warm-up pass outside the timed window: first launch pays module load / context init (main.cpp GPU warm-up pass)
    be.BeginSession(job.scene, render, job.wavelengths[0], 1024)
    be.TraceLayer(1024)
    be.EndSession()
    be.ReadbackXyzAccum()
    if job.color_classes:
        be.ReadbackClassLanes()   # the warm-up rays must not stay in the lanes
    setup = time.perf_counter() - t0
    t1 = time.perf_counter()
    rays = 0
    for wl in job.wavelengths:
        left = per_wl
        while left > 0:
            n = min(left, DISPATCH_RAYS if job.scene.layer_count == 1 else DISPATCH_RAYS_MULTI)
            be.BeginSession(job.scene, render, wl, n)
            for li in range(job.scene.layer_count):
                be.TraceLayer(n if li == 0 else 0)
                if li + 1 < job.scene.layer_count:
                    be.Recombine(True)
            be.EndSession()
            left -= n
            rays += n
        be.ConsumeDeviceFused()  # drain window
        if progress:
            progress(rays)
    be.sync()
    active = time.perf_counter() - t1
    return {"rays": rays, "setup_sec": setup, "active_sec": active, "backend": be, "render": render, "render_id": rid}


def write_ppm(path, rgb):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(rgb.tobytes())


def write_image(path, rgb, fmt, quality):
    """stbi_write_jpg / stbi_write_png of main.cpp:262-273 (JPEG: quality as given, no chroma subsampling above 90 like stb)"""
    from PIL import Image
    if fmt == "png":
        Image.fromarray(rgb).save(path, "PNG")
    else:
        Image.fromarray(rgb).save(path, "JPEG", quality=int(quality), subsampling=0 if quality > 90 else 2)
    print("Saved: %s (%dx%d)" % (path, rgb.shape[1], rgb.shape[0]))


def save_all_renders(job, args):
    """`-f cfg -o dir [--format png] [--quality q]`: what the reference CLI leaves in the output directory (SaveRenderResults /
    SaveCompositeResults, main.cpp:251-315): img_<id>.<fmt> per render entry, img_<id>_components.<fmt> with raypath_color."""
    if not os.path.isdir(args.output_dir):   # the reference fails on an output directory that does not exist (test_errors.py: nonexistent output dir)
        print("Error: output directory does not exist: %s" % args.output_dir, file=sys.stderr)
        return 2
    if not job.renders:
        raise config.ConfigError("config has no render entry")
    rc = 0
    for rid in sorted(job.renders):
        try:
            res = run_job(job, rid, args.seed, args.device, args.max_rays)
        except BackendUnavailableError as e:
            print("backend unavailable: %s" % e, file=sys.stderr)
            return 3
        be = res["backend"]
        meta = job.render_meta.get(rid, {})
        rgb, _, total_intensity = be.Snapshot(intensity_factor=meta.get("intensity_factor", 1.0), ray_color=meta.get("ray_color", (-1.0, -1.0, -1.0)),
                                              background=meta.get("background", (0.0, 0.0, 0.0)), want_xyz=False)
        write_image(os.path.join(args.output_dir, "img_%02d.%s" % (rid, args.format)), rgb, args.format, args.quality)
        if job.color_classes:
            ok, _, srgb, p99 = be.CompositeColorClasses(job.color_meta, job.color_mode, 2.0 ** args.display_ev, meta.get("intensity_factor", 1.0))
            if ok:
                write_image(os.path.join(args.output_dir, "img_%02d_components.%s" % (rid, args.format)), srgb, args.format, args.quality)
        be.close()
    return rc


def main(argv=None):
    ap = argparse.ArgumentParser(prog="ice_halo_sim_amd.cli")
    ap.add_argument("-f", "--config", required=True, help="Lumice JSON configuration file")
    ap.add_argument("--render", type=int, default=None, help="render id to trace (default: lowest id)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--max-rays", type=int, default=None, help="cap the total root rays (required for ray_num: infinite)")
    ap.add_argument("--benchmark", action="store_true", help="print the [BENCHMARK] JSON line")
    ap.add_argument("-o", "--output-dir", default=None, help="the reference CLI's output contract (main.cpp:196-315): write img_<id>.<fmt> for EVERY "
                    "render entry of the config (one run per entry: the seam carries one renderer) and, for raypath_color configs, img_<id>_components.<fmt>")
    ap.add_argument("--format", default="jpg", choices=("jpg", "png"), help="image format of -o (default jpg)")
    ap.add_argument("--quality", type=int, default=95, help="JPEG quality of -o (default 95, 4:4:4 like stb_image_write above 90)")
    ap.add_argument("--backend", default="hip", help="accepted for the reference's command lines (auto | cpu | metal | cuda | hip); this engine has one backend")
    ap.add_argument("--out-rgb", default=None, help="write the sRGB image as binary PPM")
    ap.add_argument("--out-xyz", default=None, help="write the raw XYZ snapshot as .npy")
    ap.add_argument("--out-lanes", default=None, help="raypath_color configs: write the per-class Y lanes (classes, H, W) as .npy")
    ap.add_argument("--out-composite", default=None, help="raypath_color configs: write the class composite (the config's mode: dominant | additive | "
                    "painter; component_compositor.cpp) as binary PPM, composited on the device")
    ap.add_argument("--display-ev", type=float, default=0.0, help="display-time EV of the composite (display_exposure_scale = 2^EV)")
    args = ap.parse_args(argv)
    if not 1 <= args.quality <= 100:
        print("Error: --quality must be between 1 and 100, got %d" % args.quality, file=sys.stderr)
        return 2
    try:
        job = config.load_config(args.config)
        for w in getattr(job, "warnings", []):
            print("[warning] " + w)
        if job.ray_num is None and args.max_rays is None:
            raise config.ConfigError('ray_num is "infinite": pass --max-rays')
        if args.output_dir is not None and not args.benchmark and args.render is None:
            return save_all_renders(job, args)
        if args.output_dir is not None:
            print("[warning] -o / --output-dir is ignored with --render / --benchmark (nothing is saved)", file=sys.stderr)
        wall0 = time.perf_counter()
        res = run_job(job, args.render, args.seed, args.device, args.max_rays)
    except BackendUnavailableError as e:
        print("backend unavailable: %s" % e, file=sys.stderr)
        return 3
    except config.ConfigError as e:
        print("config error: %s" % e, file=sys.stderr)
        return 2
    be = res["backend"]
    meta = job.render_meta.get(res["render_id"], {})
    rgb, xyz, total_intensity = be.Snapshot(intensity_factor=meta.get("intensity_factor", 1.0), ray_color=meta.get("ray_color", (-1.0, -1.0, -1.0)),
                                            background=meta.get("background", (0.0, 0.0, 0.0)))
    wall = time.perf_counter() - wall0
    if args.out_rgb:
        write_ppm(args.out_rgb, rgb)
    if args.out_xyz:
        np.save(args.out_xyz, xyz)
    if args.out_composite and job.color_classes:   # before the lanes are drained: the composite reads them where they are
        ok, _, srgb, p99 = be.CompositeColorClasses(job.color_meta, job.color_mode, 2.0 ** args.display_ev, meta.get("intensity_factor", 1.0))
        if ok:
            write_ppm(args.out_composite, srgb)
        else:
            print("no composite: the participating classes hold no energy (P99 %.3g)" % p99, file=sys.stderr)
    if args.out_lanes and job.color_classes:
        np.save(args.out_lanes, be.ReadbackClassLanes())
    if args.benchmark:
        active = res["active_sec"]
        basis = "steady" if active >= 0.05 else "active_short"
        out = {"mode": "multi", "workers": 1, "cores": os.cpu_count(), "rays": res["rays"], "wall_sec": round(wall, 3),
               "setup_sec": round(res["setup_sec"], 3), "active_sec": round(active, 3),
               "rays_per_sec": round(res["rays"] / max(active, 1e-9), 1), "rate_basis": basis, "backend": "hip",
               "resolution": [res["render"].width, res["render"].height]}
        print("[BENCHMARK] " + json.dumps(out))
    else:
        print("traced %d root rays in %.3f s (setup %.3f s); landed intensity %.6g" % (res["rays"], res["active_sec"], res["setup_sec"], total_intensity))
    be.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
