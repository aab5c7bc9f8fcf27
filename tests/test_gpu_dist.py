"""The N > 1 path on the GPU box's ONE device: two ranks share it over gloo and run exactly what bench.py runs per rank —
ice_halo_sim_amd.dist.ShardedTracer (the HIP backend accumulating straight into a torch tensor, rank-offset RNG counters) and the
drain-point reduce — so the reduced image can be checked against the ranks' own images.  (The driver's 8-GPU run uses RCCL on the
same code; gloo reduces CUDA tensors as all_reduce, which dist.reduce_image knows.)"""
import os
import socket

import numpy as np
import pytest

from ice_halo_sim_amd import abi, scenes

pytestmark = pytest.mark.gpu

W, H, N = 480, 270, 3 << 20   # per rank: above the hit log's 2 Mi-ray threshold, so the production route runs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from ice_halo_sim_amd.dist import ShardedTracer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, rd = scenes.config2_scene(), scenes.config2_render(W, H)
    tr = ShardedTracer(sc, rd, seed=42, device=0, rank=rank, world=world, **{"async": 1})
    for wl in (scenes.wl_discrete(530.0), scenes.wl_discrete(610.0)):
        tr.trace_session_layers(wl, N)
    tr.backend.flush()                                         # (ShardedTracer defers the closing folds: the tensor is current after this)
    torch.cuda.synchronize()
    own = tr.acc[: W * H * 3].cpu().numpy().copy()          # this rank's image before the reduce
    own_landed = tr.backend.take_landed()
    tr.landed = own_landed                                     # (readback adds the device tally: put it back)
    route = tr.backend.last_route()
    tr.reduce_to_root()
    img, landed = tr.readback()
    np.save(os.path.join(out_dir, "own%d.npy" % rank), own)
    np.save(os.path.join(out_dir, "red%d.npy" % rank), img.ravel())
    np.save(os.path.join(out_dir, "meta%d.npy" % rank), np.array([own_landed, landed, route.mode_mask, route.accum_mask], np.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reduce_to_the_sum_of_their_images(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    own = [np.load(tmp_path / ("own%d.npy" % r)) for r in range(2)]
    red = [np.load(tmp_path / ("red%d.npy" % r)) for r in range(2)]
    meta = [np.load(tmp_path / ("meta%d.npy" % r)) for r in range(2)]
    assert all(int(m[2]) == abi.MODE_PLAIN and int(m[3]) == abi.ACCUM_LOG for m in meta)      # the production route, on both ranks
    # rank 0 holds the sum of the two images (one float add per element: exact), rank 1 is drained
    assert np.array_equal(red[0], own[0] + own[1])
    assert not red[1].any()
    assert meta[0][1] == pytest.approx(meta[0][0] + meta[1][0], rel=1e-12) and meta[1][1] == 0.0
    # the two shards are different rays (rank-offset counters), equally bright
    assert not np.allclose(own[0], own[1])
    assert meta[0][0] == pytest.approx(meta[1][0], rel=5e-3)
    assert own[0].sum(dtype=np.float64) == pytest.approx(own[1].sum(dtype=np.float64), rel=5e-3)


# ---- the drain off the critical path: several steps, reduce on the side stream into alternating tensors == reduce in line ---------------------
def _worker_steps(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from ice_halo_sim_amd.dist import ShardedTracer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, rd = scenes.config2_scene(), scenes.config2_render(W, H)
    steps = [(scenes.wl_discrete(530.0), scenes.wl_discrete(610.0 + 10.0 * k)) for k in range(5)]   # an odd count: the root's light ends up spread over BOTH tensors
    out = {}
    for tag, overlap in (("side", True), ("inline", False)):
        tr = ShardedTracer(sc, rd, seed=42, device=0, rank=rank, world=world, overlap_reduce=overlap, **{"async": 1})
        assert tr.two == overlap and len(tr.accs) == (2 if overlap else 1)
        for wls in steps:
            for wl in wls:
                tr.trace_session_layers(wl, N)
            tr.reduce_to_root()                                  # nothing is read between the steps
        assert len(tr.reduce_log) == 5 and tr.cur == (1 if overlap else 0)
        img, landed = tr.readback()
        assert not any(a.any().item() for a in tr.accs)         # readback leaves every tensor drained
        out[tag] = (img, landed)
        tr.backend.close()
    # the truth both are held to: every rank's shard traced alone (no collective anywhere near it), the images added on the host
    truth = np.zeros(W * H * 3, np.float64)
    for r in range(world):
        t1 = ShardedTracer(sc, rd, seed=42, device=0, rank=r, world=1, **{"async": 1})
        for wls in steps:
            for wl in wls:
                t1.trace_session_layers(wl, N)
        t1.backend.flush()
        torch.cuda.synchronize()
        truth += t1.acc[: W * H * 3].cpu().numpy().astype(np.float64)
        t1.backend.close()
    np.save(os.path.join(out_dir, "truth%d.npy" % rank), truth)
    np.save(os.path.join(out_dir, "side%d.npy" % rank), out["side"][0].ravel())
    np.save(os.path.join(out_dir, "inline%d.npy" % rank), out["inline"][0].ravel())
    np.save(os.path.join(out_dir, "landed%d.npy" % rank), np.array([out["side"][1], out["inline"][1]], np.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_drain_on_the_side_stream_equals_the_in_line_drain_over_several_steps(tmp_path):
    """dist.ShardedTracer's two accumulator tensors (round 6): five steps of two sessions each, every step closed by reduce_to_root with
    nothing read in between — once with the collective queued on the side stream while the next step traces into the other tensor, once
    serialised (the trace stream waits for every collective).  Both are held to the TRUTH: each rank's shard traced alone, with no
    collective, and the images added on the host.  Same rays (rank-offset counters, same seed), so the root's image is that sum taken in
    another order, the landed weight is equal, and the other rank holds nothing."""
    import torch.multiprocessing as mp
    mp.spawn(_worker_steps, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    side = [np.load(tmp_path / ("side%d.npy" % r)) for r in range(2)]
    inline = [np.load(tmp_path / ("inline%d.npy" % r)) for r in range(2)]
    landed = [np.load(tmp_path / ("landed%d.npy" % r)) for r in range(2)]
    truth = np.load(tmp_path / "truth0.npy")
    assert truth.sum() > 0
    for name, img in (("side stream", side[0]), ("serialised", inline[0])):
        assert np.abs(img - truth).max() <= 2e-6 * float(truth.max()), name       # float sums in another order
        assert img.sum(dtype=np.float64) == pytest.approx(truth.sum(), rel=1e-6), name
    assert landed[0][0] == pytest.approx(landed[0][1], rel=1e-9) and landed[0][0] > 0
    assert not side[1].any() and not inline[1].any() and landed[1][0] == 0.0 and landed[1][1] == 0.0


# ---- raypath-colour job over two ranks: lanes reduced, ONE composite on the root's device (ShardedTracer.composite) ---------------------
def _colour_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from ice_halo_sim_amd.dist import ShardedTracer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = scenes.filter_term
    ee = lambda lo=1, hi=None: T("entry_exit", min_len=lo, max_len=hi)
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    ax = scenes.axis(zenith={"type": "gauss", "mean": 90.0, "std": 20.0}, azimuth=full, roll=full)
    sc = scenes.scene([(0.0, [scenes.entry(scenes.prism_crystal(1.3), ax, 0.5, 1, color_id=1), scenes.entry(scenes.prism_crystal(0.4), ax, 0.5, 2, color_id=2)])],
                      max_hits=6, sun_altitude=25.0)
    rd = scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 256, 128, visible=abi.VISIBLE_FULL)
    tr = ShardedTracer(sc, rd, seed=42, device=0, rank=rank, world=world)
    tr.backend.set_color([scenes.color_set([(ee(1), "", 0)]), scenes.color_set([(ee(2, 2), "", 1), (ee(3), "", 2)])],
                         [scenes.color_class([0]), scenes.color_class([1]), scenes.color_class([2])])
    tr.trace_session_layers(scenes.wl_discrete(550.0), 1 << 20)
    torch.cuda.synchronize()
    # this rank's own lanes, read without draining: composite() reads them back itself, so take a copy through a second readback + reload
    own = tr.backend.ReadbackClassLanes()
    tr.backend.LoadClassLanes(own)
    disp = [{"color": c} for c in ([1, 0, 0], [0, 1, 0], [0, 0, 1])]
    res = tr.composite(disp, "painter")
    assert (res is None) == (rank != 0)
    np.save(os.path.join(out_dir, "lanes%d.npy" % rank), own)
    np.save(os.path.join(out_dir, "landed%d.npy" % rank), np.array([tr.landed]))
    if rank == 0:
        ok, lin, srgb, p99 = res
        assert ok
        np.save(os.path.join(out_dir, "lin.npy"), lin)
        np.save(os.path.join(out_dir, "p99.npy"), np.array([p99], np.float32))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_composite_their_summed_lanes_once_on_the_root(tmp_path):
    import torch.multiprocessing as mp
    from tests.test_compositor import oracle_composite
    mp.spawn(_colour_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    lanes = [np.load(tmp_path / ("lanes%d.npy" % r)) for r in range(2)]
    landed = [float(np.load(tmp_path / ("landed%d.npy" % r))[0]) for r in range(2)]
    lin, p99 = np.load(tmp_path / "lin.npy"), float(np.load(tmp_path / "p99.npy")[0])
    assert not np.array_equal(lanes[0], lanes[1]) and landed[0] == pytest.approx(landed[1], rel=2e-2)
    cls = [{"color": c, "bits": 1 << i} for i, c in enumerate(([1, 0, 0], [0, 1, 0], [0, 0, 1]))]
    ok, want, _, want_p99 = oracle_composite(lanes[0] + lanes[1], landed[0] + landed[1], cls, "painter", 1.0, 1.0)
    assert ok and want_p99 == p99
    assert np.array_equal(lin.reshape(-1, 3), want)


# ---- `python bench.py --gpus N` started WITHOUT a launcher must launch its own ranks (round-4 review: it used to SystemExit) -------------
def test_bench_launches_its_own_ranks_when_started_plainly():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HALO_BENCH_BACKEND"] = "gloo"            # two ranks on this box's one device; the driver's 8-GPU run is the same code over RCCL
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--repeats", "1",
                        "--rays-per-wl", "3000000", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["rays_per_step_per_gpu"] == 9 * 3_000_000
    mg = out["multi_gpu"]
    assert mg["ranks_seen"] == 2 and mg["backend"] == "gloo"
    assert all(mg["check"][k] for k in ("rank_images_differ", "rank_energies_within_1pct", "reduced_image_is_the_sum", "nonroot_ranks_drained"))
