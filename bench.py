#!/usr/bin/env python3
"""bench.py — rays/s of the MI355X trace hot path on BASELINE.json's headline workload.

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one rank per GPU, RCCL).  One "step" = one pass of the hot path over one batch: configs[1] — single-scatter
hex column (examples/config_example.json crystal 3), 9 wavelengths x 50 M root rays, fisheye_equal_area
1920x1080 — i.e. 450 M root rays per GPU per step.  Rays shard by index range (disjoint RNG counter ranges per
rank, no data-path collective); every step ends with ONE RCCL sum-reduce of the W*H*3(+4) fp32 accumulator to
rank 0 (the reference's drain point, simulator.cpp:1409-1477).  Per-GPU work is fixed → "scaling": "weak".

Rank 0 prints ONE JSON line.  `roofline` prices the fused kernel against HBM (no dense contraction → no MFMA):
achieved = algorithmic accumulator bytes (in-frame pixel writes x 3 channels x 4 B x read+write)
per launch / mean launch duration from HIP events on the launch stream.  `cpu_baseline` times the CPU oracle
(a port of the reference's algorithm; the reference itself cannot be built here) on the host cores on a bounded
sample of the same workload — a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); the launcher
# exports it already — keep it if this file is started some other way
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(rays_per_wl_hint, budget_s=12.0):
    """CPU oracle ("port") on all host cores, bounded to ~budget_s of work on the same workload shape."""
    from ice_halo_sim_amd import scenes
    from tests._oracle_backend import OracleBackend, run_session
    cores = os.cpu_count() or 1
    threads = min(cores, 128)
    sc, rd = scenes.config2_scene(), scenes.config2_render()
    ob = OracleBackend(seed=42, threads=threads)
    t0 = time.perf_counter()
    run_session(ob, sc, rd, scenes.wl_discrete(550.0), 100_000)  # calibration (also warms the LUT/page cache)
    rate = 100_000 / max(time.perf_counter() - t0, 1e-6)
    per_wl = int(max(20_000, min(rays_per_wl_hint, rate * budget_s / 9)))
    t0 = time.perf_counter()
    for wl in scenes.CONFIG_WAVELENGTHS_9:
        run_session(ob, sc, rd, scenes.wl_discrete(wl), per_wl)
    dt = time.perf_counter() - t0
    ob.close()
    return {"value": 9 * per_wl / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": "configs[1] shape, 9 wavelengths x %d rays (%.1f s of CPU work), OpenMP over rays" % (per_wl, dt)}


def _pmc_mean(name, key):
    """mean-per-dispatch value of counter `key` for the trace kernel in a committed rocprofv3 PMC summary (tools/rocpd_summary.py)"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
    if not os.path.exists(path):
        return None
    val = None
    for line in open(path):
        if "halo_trace_kernel" in line and (" " + key + " ") in line:
            val = float(line.split(" " + key + " ")[1].split()[0])
    return val


def pmc_traffic_per_launch():
    """HBM-side bytes per trace-kernel launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_bench_pmc_{fetch,write}_size.txt; counters are collected in their own runs, never inside the timed one).
    FETCH_SIZE / WRITE_SIZE are in KB; this kernel's traffic is scattered atomics, for which the guide calls the counters
    uncalibrated — reported as measured, uncorrected."""
    f, w = _pmc_mean("r01_bench_pmc_fetch_size.txt", "FETCH_SIZE"), _pmc_mean("r01_bench_pmc_write_size.txt", "WRITE_SIZE")
    return None if f is None or w is None else (f + w) * 1024.0


def pmc_valu(rays_per_launch):
    """VALU issue utilisation and instructions per 64-ray wave pass of the trace kernel, from the committed PMC pass
    (profiles/r01_bench_pmc_insts.txt: per-dispatch means are per counter instance = 32 SIMDs).  Issue fraction =
    VALU wave-instructions per SIMD x 4 cycles (a wave64 VALU instruction occupies a SIMD for 4 cycles) / kernel duration
    of that same pass at the nominal 2.4 GHz."""
    insts = _pmc_mean("r01_bench_pmc_insts.txt", "SQ_INSTS_VALU")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_bench_pmc_insts.txt")
    avg_us = None
    if os.path.exists(path):
        for line in open(path):
            if "halo_trace_kernel" in line and "SQ_" not in line:
                avg_us = float(line.split()[-4])          # calls total_us avg_us min_us max_us pct
                break
    if not insts or not avg_us:
        return None
    return {"valu_issue_frac": (insts / 32.0) * 4.0 / (avg_us * 1e-6 * 2.4e9), "valu_insts_per_wave_ray": insts * 32.0 / (rays_per_launch / 64.0),
            "assumes": "4 cycles per wave64 VALU instruction, 2.4 GHz", "source": "profiles/r01_bench_pmc_insts.txt"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rays-per-wl", type=int, default=50_000_000, help="root rays per wavelength per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--aggregate", type=int, default=-1)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
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

    from ice_halo_sim_amd import scenes
    from ice_halo_sim_amd.backend import HipTraceBackend
    from ice_halo_sim_amd.dist import ShardedTracer

    sc, rd = scenes.config2_scene(), scenes.config2_render()
    tracer = ShardedTracer(sc, rd, seed=42, device=local_rank, rank=rank, world=world, **{"async": 1})
    if args.blocks_per_cu > 0:
        tracer.backend.set_option("blocks_per_cu", args.blocks_per_cu)
    if args.aggregate >= 0:
        tracer.backend.set_option("aggregate", args.aggregate)
    wls = [scenes.wl_discrete(w) for w in scenes.CONFIG_WAVELENGTHS_9]
    n = args.rays_per_wl

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        stats = []
        for wl in wls:
            stats.append(tracer.trace_session(wl, n))      # this rank's shard of the batch
        tracer.reduce_to_root()                              # one RCCL sum-reduce at the drain point
        return stats

    for _ in range(args.warmup):
        step()
    tracer.zero()
    tracer.backend.collect_stats()                            # drop the warm-up tallies
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()                                               # dispatches are queued; nothing waits on the host per launch
    barrier()
    dt = time.perf_counter() - t0
    st = tracer.backend.collect_stats()                      # HIP-event kernel times + device tallies of the timed region
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    rays_per_rank = args.steps * len(wls) * n
    total_rays = rays_per_rank * world
    launches, kernel_ms, pixel_hits, exits = int(st.launches), float(st.kernel_ms), int(st.pixel_hits), int(st.exit_count)
    assert int(st.root_count) == rays_per_rank

    img, landed = tracer.readback()                          # collective: landed-weight scalars are summed here, once
    if rank == 0:
        avg_launch_s = kernel_ms * 1e-3 / max(launches, 1)
        alg_bytes_per_launch = pixel_hits * 24.0 / max(launches, 1)   # 3 ch x 4 B x (read + write) per in-frame pixel hit
        achieved = alg_bytes_per_launch / max(avg_launch_s, 1e-12) / 1e9
        out = {
            "metric": "rays/sec (whole node) at 9 wavelengths, single-scatter hex column",
            "value": total_rays / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: single-scatter hex column (prism h=1.3, zenith gauss(90,0.3)), 9 wavelengths x %d root rays per GPU per step, max_hits 7, fisheye_equal_area fov 180 1920x1080 visible upper" % n,
                       "rays_per_step_per_gpu": len(wls) * n, "resolution": [rd.width, rd.height],
                       "sharding": "root-ray index ranges, 1 RCCL reduce of %d floats per step" % (rd.width * rd.height * 3 + 4),
                       "exits_per_root": exits / max(rays_per_rank, 1), "landed_weight_rank0_image": landed},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_per_launch(),
                         "traffic_source": "profiles/r01_bench_pmc_{fetch,write}_size.txt (separate rocprofv3 --pmc passes of this command, bytes per launch, uncorrected)",
                         "kernel": "halo_trace_kernel<0,0,true,false>", "launches": launches,
                         "avg_launch_ms": avg_launch_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "kernel_rays_per_s": (rays_per_rank / max(kernel_ms * 1e-3, 1e-12)),
                         "valu": pmc_valu(n),
                         "note": "fused kernel keeps rays in registers: HBM sees only accumulator RMWs, so the path is VALU-issue-bound, not HBM-bound (see `valu`; DESIGN.md §4)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
